import importlib
for m in ["numpy", "scipy", "torch", "tqdm", "yaml", "PIL", "matplotlib", "huggingface_hub", "pytest", "psutil", "h5py", "gymnasium", "requests",
          "hypothesis", "pytest_timeout", "xdist"]:
    try:
        importlib.import_module(m)
        print(m, "OK")
    except Exception as e:
        print(m, "MISSING", type(e).__name__)
