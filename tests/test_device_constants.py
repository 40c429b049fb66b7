"""maniskill_amd.fused_step.DeviceConstants / _MaskedSelection / graph_safety on plain torch tensors (no env, no reference checkout): the lazy ``y[mask]`` has
to be either exactly what eager torch computes or a loud refusal -- never a silently different value (round 5's advisor findings, each reproduced here)."""
import pytest
import torch

from maniskill_amd.fused_step import DeviceConstants, Unsupported, _verdict, graph_safety


def _mode():
    return DeviceConstants("cpu")


def test_masked_copy_idiom_is_a_select_with_eager_values():
    x, y = torch.arange(6.0), torch.arange(6.0) * 10
    m = torch.tensor([True, False, True, False, False, True])
    ex = x.clone()
    ex[m] = y[m]
    with _mode() as c:
        x[m] = y[m]
    assert torch.equal(x, ex) and c.rewritten == 1


def test_save_restore_idiom_is_refused_not_aliased():
    """old = x[m]; x[m] = 0; y[m] = old: eager torch restores the OLD values; the lazy selection would read the zeros."""
    x, y = torch.arange(3.0) + 1, torch.zeros(3)
    m = torch.tensor([True, False, True])
    zero = torch.zeros(())
    with pytest.raises(Unsupported):
        with _mode():
            old = x[m]
            x[m] = zero
            y[m] = old


def test_selection_used_after_its_source_was_written_is_refused():
    """keep = x[m]; x += 10; x[m] = keep"""
    x = torch.arange(3.0)
    m = torch.tensor([True, False, True])
    with pytest.raises(Unsupported):
        with _mode():
            keep = x[m]
            x += 10
            x[m] = keep


def test_selection_made_real_after_a_write_is_refused_as_well():
    x = torch.arange(4.0)
    m = torch.tensor([True, False, True, False])
    with pytest.raises(Unsupported):
        with _mode():
            sel = x[m]
            x.mul_(2)
            sel.sum()          # any other use makes the selection real: it must be the values at selection time


def test_selection_arithmetic_keeps_eager_values():
    r, b = torch.arange(5.0), torch.arange(5.0) + 0.5
    m = torch.tensor([False, True, True, False, True])
    er = r.clone()
    er[m] += b[m]
    with _mode():
        r[m] += b[m]
    assert torch.equal(r, er)


def test_inverted_mask_cache_sees_in_place_edits_of_its_result():
    """nm = ~m; nm[1] = False; x[~m] = ...: the second ~m is a fresh inversion in eager torch"""
    x = torch.zeros(3)
    m = torch.tensor([True, False, False])
    one = torch.ones(())
    with _mode():
        nm = ~m
        nm[1] = False
        x[~m] = one
    assert x.tolist() == [0.0, 1.0, 1.0]


def test_device_constants_are_reentrant_and_put_synchronize_back():
    orig = torch.cuda.synchronize
    c = _mode()
    with c:
        with c:
            assert torch.cuda.synchronize is not orig
        assert torch.cuda.synchronize is not orig      # the inner exit must not restore (or lose) the original
    assert torch.cuda.synchronize is orig


def test_the_watch_takes_index_select_with_a_dim():
    x, idx = torch.arange(12.0).view(3, 4), torch.tensor([0, 2])

    def step(a):
        return torch.index_select(x, 1, idx) + a
    v = graph_safety(step, torch.zeros(()))
    assert v["sync"] == []


def test_the_watch_still_flags_mask_indexing():
    x = torch.arange(4.0)

    def step(a):
        return x[x > 1.0] + a
    v = graph_safety(step, torch.zeros(()))
    assert any("mask" in s or "nonzero" in s for s in v["sync"])


def test_an_error_inside_the_watch_becomes_unsupported():
    def step(a):
        raise TypeError("'int' object is not iterable")
    with pytest.raises(Unsupported):
        _verdict(step, torch.zeros(()))


def test_selections_through_a_mask_known_to_be_all_true_are_the_whole_source():
    """actor.py:389: ``buf[idx[reset_mask[scene_idxs]], :7] = pose`` outside a reset: the mask is all True, so nothing needs nonzero()"""
    reset_mask = torch.ones(4, dtype=torch.bool)
    scene_idxs = torch.tensor([0, 1, 2, 3])
    idx = torch.tensor([2, 5, 8, 11])
    buf, pose = torch.zeros(12, 3), torch.arange(12.0).view(4, 3)
    ebuf = buf.clone()
    ebuf[idx[reset_mask[scene_idxs]], :3] = pose
    c = _mode()
    c.all_true = lambda: reset_mask
    with c:
        sel = idx[reset_mask[scene_idxs]]
        assert type(sel) is torch.Tensor and sel.data_ptr() != idx.data_ptr()      # a plain copy, as eager indexing makes one
        buf[sel, :3] = pose
    assert torch.equal(buf, ebuf)


def test_a_mask_that_is_not_all_true_takes_the_ordinary_path():
    reset_mask = torch.tensor([True, False, True])
    x = torch.arange(3.0)
    c = _mode()
    c.all_true = lambda: reset_mask
    with c:
        out = x[reset_mask] * 1.0
    assert out.tolist() == [0.0, 2.0]


def test_a_mask_with_a_slice_behind_it_is_the_mask_over_the_view():
    """drawing/draw.py:178-183: ``pos[touching, :2] = tcp[touching, :2]; pos[touching, 2] = height`` -- no nonzero(), the same values"""
    m = torch.tensor([True, False, True, False])
    tcp = torch.arange(12.0).view(4, 3)
    pos, epos = torch.zeros(4, 3), torch.zeros(4, 3)
    epos[m, :2] = tcp[m, :2]
    epos[m, 2] = 0.5
    c = _mode()
    with c:
        pos[m, :2] = tcp[m, :2]
        pos[m, 2] = 0.5
    assert torch.equal(pos, epos) and c.rewritten == 2

    def step(a):
        with c:
            out = torch.zeros(4, 3)
            out[m, :2] = tcp[m, :2] * 2.0
            out[:, 0] += a
            out[m, 2] = 0.5
            return out
    v = graph_safety(step, torch.zeros(()))
    assert v["sync"] == [] and v["flow"] == [], v


def test_a_row_picked_per_env_through_a_one_hot_mask_is_a_gather():
    """rotation_conversions.py:161-163 (matrix_to_quaternion): ``cand[F.one_hot(q_abs.argmax(-1), 4) > 0.5, :]`` -- boolean indexing sized by data; here the gather"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    cand, q_abs = torch.rand(6, 4, 4, generator=g), torch.rand(6, 4, generator=g)
    want = cand[F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :].reshape(6, 4)
    c = _mode()

    def step(a):
        with c:
            return cand[F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :].reshape(6, 4) + a
    assert torch.equal(step(torch.zeros(())), want)
    v = graph_safety(step, torch.zeros(()))
    assert v["sync"] == [] and v["flow"] == [], v


def test_normalising_the_rows_a_mask_names_needs_no_nonzero():
    """sapien_utils.py:349-355 (look_at's normalize_tensor): ``x[zero] = torch.zeros(3); x[~zero] /= norm[~zero].view(-1, 1)``"""
    x0 = torch.tensor([[3.0, 0.0, 4.0], [0.0, 0.0, 0.0], [0.0, 2.0, 0.0]])

    def normalize(x):
        norm = torch.linalg.norm(x, dim=-1)
        zero = norm < 1e-6
        x[zero] = torch.zeros(3)
        x[~zero] /= norm[~zero].view(-1, 1)
        return x
    want = normalize(x0.clone())
    c = _mode()

    def step(a):
        with c:
            return normalize(x0.clone() + a)
    assert torch.equal(step(torch.zeros(())), want)
    v = graph_safety(step, torch.zeros(()))
    assert v["sync"] == [] and v["flow"] == [], v
