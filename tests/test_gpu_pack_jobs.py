"""dmd_pack_jobs / engine.PackCache: every kernel-layout copy of the convolution parameters in one launch -- bitwise the
copies the single-tensor entry points (dmd_pack_conv_weight, dmd_pack_conv_weight_f16x2) build from the torch-transformed
weights, including the transposed / sliced / zero-padded weights of the data gradient (autograd of F.conv2d,
/root/reference/src/trainer.py:366); refreshed in place when a parameter changes."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _transposed(w, c0, c1, cin_pad_to=0):
    wt = w.detach()[:, c0:c1].flip(2, 3).transpose(0, 1).contiguous()
    if cin_pad_to > wt.shape[1]:
        wt = torch.cat((wt, torch.zeros(wt.shape[0], cin_pad_to - wt.shape[1], *wt.shape[2:], device=wt.device)), 1)
    return wt


def test_every_kind_of_copy_matches_the_single_tensor_packs_and_refreshes_in_place():
    from diamond_amd import engine as E, native as nv

    torch.manual_seed(0)
    convs = {
        "c64": nn.Conv2d(64, 64, 3, padding=1), "cat128": nn.Conv2d(128, 64, 3, padding=1), "c32": nn.Conv2d(32, 32, 3, padding=1),
        "proj": nn.Conv2d(128, 64, 1), "in": nn.Conv2d(15, 64, 3, padding=1), "out": nn.Conv2d(64, 3, 3, padding=1),
        "down": nn.Conv2d(64, 64, 3, stride=2, padding=1), "qkv": nn.Conv2d(64, 192, 1),
    }
    for c in convs.values():
        c.to(DEV)
    cache = E.PackCache()

    def expect():
        out = {}
        for k, c in convs.items():
            out[k, "w"] = nv.pack_conv_weight(c.weight)
            if k in ("c64", "cat128", "c32", "proj"):
                out[k, "w16"] = nv.pack_conv_weight_f16x2(c.weight)
        wp = torch.zeros(32, 64, 3, 3, device=DEV)
        wp[:3] = convs["out"].weight.detach()
        out["out", "head16"] = nv.pack_conv_weight_f16x2(wp)
        out["out", "w32"] = nv.pack_conv_weight(convs["out"].weight, 32)
        out["out", "b32"] = nv.pad_vector(convs["out"].bias, 32)
        out["c64", "d"] = nv.pack_conv_weight(_transposed(convs["c64"].weight, 0, 64))
        out["c64", "d16"] = nv.pack_conv_weight_f16x2(_transposed(convs["c64"].weight, 0, 64))
        out["cat128", "d_hi"] = nv.pack_conv_weight(_transposed(convs["cat128"].weight, 64, 128))
        out["cat128", "d16_hi"] = nv.pack_conv_weight_f16x2(_transposed(convs["cat128"].weight, 64, 128))
        out["out", "d_pad16"] = nv.pack_conv_weight(_transposed(convs["out"].weight, 0, 64, 16))
        out["out", "d16_pad16"] = nv.pack_conv_weight_f16x2(_transposed(convs["out"].weight, 0, 64, 16))
        out["in", "d"] = nv.pack_conv_weight(_transposed(convs["in"].weight, 0, 15))
        return out

    def mine():
        out = {}
        for k, c in convs.items():
            out[k, "w"] = cache.conv_weight(c)
            if k in ("c64", "cat128", "c32", "proj"):
                out[k, "w16"] = cache.conv_weight_f16x2(c)
        out["out", "head16"] = cache.conv_weight_f16x2_head(convs["out"])
        out["out", "w32"] = cache.conv_weight(convs["out"], 32)
        out["out", "b32"] = cache.conv_bias(convs["out"], 32)
        out["c64", "d"] = cache.dgrad_weight(convs["c64"], 0, 64)
        out["c64", "d16"] = cache.dgrad_weight(convs["c64"], 0, 64, f16x2=True)
        out["cat128", "d_hi"] = cache.dgrad_weight(convs["cat128"], 64, 128)
        out["cat128", "d16_hi"] = cache.dgrad_weight(convs["cat128"], 64, 128, f16x2=True)
        out["out", "d_pad16"] = cache.dgrad_weight(convs["out"], 0, 64, 16)
        out["out", "d16_pad16"] = cache.dgrad_weight(convs["out"], 0, 64, 16, f16x2=True)
        out["in", "d"] = cache.dgrad_weight(convs["in"], 0, 15)
        return out

    a, b = mine(), expect()
    assert a.keys() == b.keys()
    for k in a:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k
        assert torch.equal(a[k], b[k]), k
    assert cache.conv_bias(convs["c64"]).data_ptr() == convs["c64"].bias.data_ptr()  # nothing to pad: the parameter itself
    ptrs = {k: v.data_ptr() for k, v in a.items()}
    # an optimizer-like in-place update of ONE parameter: every copy is rebuilt by the next lookup, in place
    with torch.no_grad():
        for c in convs.values():
            c.weight.mul_(1.5)
            c.bias.add_(0.25)
    a2, b2 = mine(), expect()
    for k in a2:
        assert torch.equal(a2[k], b2[k]), k
        assert a2[k].data_ptr() == ptrs[k], f"{k}: rebuilt into a new buffer"
    # a write that does not bump the version is invisible until invalidate()
    convs["c64"].weight.data.mul_(2.0)
    assert torch.equal(cache.conv_weight(convs["c64"]), b2["c64", "w"])
    cache.invalidate()
    assert torch.equal(cache.conv_weight(convs["c64"]), nv.pack_conv_weight(convs["c64"].weight))
    assert torch.equal(cache.dgrad_weight(convs["c64"], 0, 64, f16x2=True), nv.pack_conv_weight_f16x2(_transposed(convs["c64"].weight, 0, 64)))


def test_a_swapped_parameter_storage_reaches_every_copy_of_that_parameter():
    """`p.data = other` (checkpoint surgery, `module.to(...)`): ALL copies of the parameter are rebuilt from the new storage --
    not only the one that is looked up first (the others' table rows would still name the old, possibly freed, storage and be
    stamped fresh) -- into the same buffers, and rows of parameters that died leave the table."""
    from diamond_amd import engine as E, native as nv

    torch.manual_seed(1)
    conv, other = nn.Conv2d(64, 64, 3, padding=1).to(DEV), nn.Conv2d(128, 64, 3, padding=1).to(DEV)
    cache = E.PackCache()
    look = lambda: (cache.conv_weight(conv), cache.conv_weight_f16x2(conv), cache.dgrad_weight(conv, 0, 64, f16x2=True),
                    cache.conv_weight(other), cache.conv_weight_f16x2(other))
    before = look()
    ptrs = [t.data_ptr() for t in before]
    frees = cache.frees_epoch
    old_storage = conv.weight.data
    conv.weight.data = torch.randn_like(conv.weight.data)  # a new storage, same version
    old_storage.fill_(float("nan"))  # whoever still reads the old storage shows
    after = look()
    assert [t.data_ptr() for t in after] == ptrs and cache.frees_epoch == frees, "copies of a swapped parameter are rebuilt in place"
    assert torch.equal(after[0], nv.pack_conv_weight(conv.weight))
    assert torch.equal(after[1], nv.pack_conv_weight_f16x2(conv.weight))
    assert torch.equal(after[2], nv.pack_conv_weight_f16x2(_transposed(conv.weight, 0, 64)))
    assert torch.equal(after[3], nv.pack_conv_weight(other.weight)) and torch.equal(after[4], nv.pack_conv_weight_f16x2(other.weight))
    # a parameter that died: its rows are dropped at the next rebuild of the table instead of being packed from freed memory
    del other, look, before, after
    import gc

    gc.collect()
    with torch.no_grad():
        conv.weight.mul_(0.5)
    assert torch.equal(cache.conv_weight_f16x2(conv), nv.pack_conv_weight_f16x2(conv.weight))
    cache.invalidate()
    assert torch.equal(cache.conv_weight(conv), nv.pack_conv_weight(conv.weight))
    assert all(e.ref() is not None for e in cache._table_entries) or cache._table is None


def test_looked_up_copies_outside_the_job_table_are_rebuilt_in_place_too():
    """PackCache.get (transposed LSTM weights, per-head qkv pieces): same buffer after an update, and after invalidate() --
    a captured graph that holds the pointer reads the new values."""
    from diamond_amd import engine as E

    w = nn.Parameter(torch.randn(48, 32, device=DEV))
    cache = E.PackCache()
    t0 = cache.get(w, "T", lambda t: t.detach().t().contiguous())
    ptr = t0.data_ptr()
    with torch.no_grad():
        w.mul_(2.0)
    t1 = cache.get(w, "T", lambda t: t.detach().t().contiguous())
    assert t1.data_ptr() == ptr and torch.equal(t1, w.detach().t())
    w.data.add_(1.0)
    cache.invalidate()
    assert cache.stale_epoch == 1
    cache.refresh()  # what a captured training step / the graphed sampler call: no lookup needed
    assert t1.data_ptr() == ptr and torch.equal(t1, w.detach().t()) and cache.frees_epoch == 0
    # a view of the parameter itself stays a view (nothing to copy, the pointer follows the parameter)
    v = cache.f32(w)
    assert v.data_ptr() == w.data_ptr()


def test_a_fused_optimizer_step_is_seen_although_it_bumps_no_version():
    """torch's fused optimizers (`AdamW(fused=True)`: one multi-tensor kernel, what GraphedTrainStep's bench line uses) update the
    parameters without touching `Tensor._version`; engine's optimizer post-step hook must mark the copies stale."""
    from diamond_amd import engine as E, native as nv
    from diamond_amd.blocks import AdaGroupNorm, FilmTable

    torch.manual_seed(3)
    net = nn.ModuleDict({"conv": nn.Conv2d(64, 64, 3, padding=1), "norm": AdaGroupNorm(64, 256)}).to(DEV)
    other = nn.Conv2d(64, 64, 3, padding=1).to(DEV)
    cache, film, cache_other = E.PackCache(), FilmTable(net), E.PackCache()
    w0, f0 = cache.conv_weight_f16x2(net["conv"]).clone(), film.weights()[0].clone()
    cache_other.conv_weight(other)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2, fused=True)
    for p in net.parameters():
        p.grad = torch.randn_like(p)
    epochs = (cache.stale_epoch, film.stale_epoch, cache_other.stale_epoch)
    opt.step()
    w1, f1 = cache.conv_weight_f16x2(net["conv"]), film.weights()[0]
    assert torch.equal(w1, nv.pack_conv_weight_f16x2(net["conv"].weight)) and not torch.equal(w1, w0)
    assert torch.equal(f1, net["norm"].linear.weight.detach()) and not torch.equal(f1, f0)
    # a graph owner watching the epochs sees the step; a cache of parameters the optimizer does not hold is left alone
    assert cache.stale_epoch == epochs[0] + 1 and film.stale_epoch == epochs[1] + 1 and cache_other.stale_epoch == epochs[2]


def _audited_setup():
    from diamond_amd import engine as E
    from diamond_amd.blocks import AdaGroupNorm, FilmTable

    torch.manual_seed(5)
    net = nn.ModuleDict({"conv": nn.Conv2d(64, 64, 3, padding=1), "conv2": nn.Conv2d(32, 32, 3, padding=1), "norm": AdaGroupNorm(64, 256),
                         "lstm": nn.LSTMCell(128, 64)}).to(DEV)
    cache, film = E.PackCache(), FilmTable(net)

    def use():
        cache.conv_weight_f16x2(net["conv"]), cache.conv_weight(net["conv2"]), cache.conv_bias(net["conv"], 128)
        cache.get(net["lstm"].weight_ih, "T", lambda w: w.detach().t().contiguous())
        cache.f32(net["lstm"].bias_ih)  # (aliases the parameter: never audited, never stale)
        film.weights()

    use()
    return E, net, cache, film, use


def _audit_all(E, wait=True):
    E.run_weight_audits()
    E.check_weight_audits(wait=wait)


@pytest.mark.parametrize("victim", ["conv.weight", "conv.bias", "lstm.weight_ih", "norm.linear.weight"])
@pytest.mark.parametrize("how", ["data.copy_", "data.mul_", "set_ under no_grad via .data view"])
def test_a_silent_parameter_write_is_detected_by_the_audit(victim, how):
    """`p.data.<op>_` changes a parameter without bumping `Tensor._version`, the stamp the packed copies are keyed on: it used to be
    a documented hole (the kernels kept computing with the old weights).  The audit (engine.WeightAudit, `dmd_checksums`)
    fingerprints what every copy was built from and compares with the live parameter: the write RAISES at the next check --
    for a convolution copy of the job table, a padded bias, a looked-up transposed LSTM weight and the FiLM table alike."""
    E, net, cache, film, use = _audited_setup()
    _audit_all(E)  # clean: nothing to report
    p = dict(net.named_parameters())[victim]
    v0 = p._version
    if how == "data.copy_":
        p.data.copy_(torch.randn_like(p))
    elif how == "data.mul_":
        p.data.mul_(1.5)
    else:
        p.data.view(-1)[3] = 7.0
    assert p._version == v0, "the write was visible after all"
    use()  # still the stale copies: no stamp changed
    with pytest.raises(RuntimeError, match="stale packed weights"):
        _audit_all(E)
    # after the documented remedy the copies are rebuilt and the audit is clean again
    cache.invalidate(), film.invalidate()
    use()
    _audit_all(E)


def test_the_audit_raises_no_false_alarm(monkeypatch):
    """visible updates (in-place ops that bump the version, a fused optimizer step that bumps none but goes through the optimizer
    hook, `p.data = other`) between the build of a copy and an audit, with and without a lookup in between: never flagged; and
    the tick path -- an audit every AUDIT_EVERY lookups, its answer read by a later lookup -- catches a silent write on its own"""
    E, net, cache, film, use = _audited_setup()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.25)
    _audit_all(E)  # copies stale BY STAMP: not the audit's business
    use()
    _audit_all(E)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2, fused=(DEV == "cuda"))
    for p in net.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    _audit_all(E)
    use()
    _audit_all(E)
    net["conv"].weight.data = torch.randn_like(net["conv"].weight)  # new storage: the stamp's pointer changes
    _audit_all(E)
    use()
    _audit_all(E)
    monkeypatch.setattr(E.WeightAudit, "AUDIT_EVERY", 8)
    for _ in range(20):
        use()
    E.check_weight_audits(wait=True)
    net["conv2"].weight.data.add_(1.0)
    with pytest.raises(RuntimeError, match="stale packed weights"):
        for _ in range(40):
            use()
            if DEV == "cuda":
                torch.cuda.synchronize()
