class ControlWindow:
    pass
