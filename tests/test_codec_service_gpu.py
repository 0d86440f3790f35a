"""Batched, hipGraph-captured codec service (clc_amd.codec) against the reference-surface model methods:
per-image streams byte-identical to model.compress(), decoded images bit-identical to model.decompress() and to the
encoder-side reconstruction, graph and eager engines identical, container round trip (also through a file)."""
import os

import pytest
import torch


def _container_items():
    return [[b"\x01\x02\x03\x04" * 5], [b"\xff" * 12]], torch.Size([4, 6])


def test_container_roundtrip(tmp_path):
    from clc_amd import codec

    strings, shape = _container_items()
    blob = codec.pack(strings, shape, (200, 300), n_refs=3, model_id=1)
    assert len(blob) == 24 + 20 + 12 and blob[:4] == b"CLC1"
    s2, sh2, meta = codec.unpack(blob)
    assert s2 == strings and tuple(sh2) == (4, 6)
    assert meta == {"image_hw": (200, 300), "n_refs": 3, "model_id": 1, "kernel_config_tag": codec.kernel_config_tag(), "same_kernel_config": True}
    n = codec.write_file(tmp_path / "a.clc", strings, shape, (200, 300), 3, 1)
    assert n == len(blob) and os.path.getsize(tmp_path / "a.clc") == n
    s3, sh3, meta3 = codec.read_file(tmp_path / "a.clc")
    assert s3 == strings and tuple(sh3) == (4, 6) and meta3 == meta
    with pytest.raises(ValueError):
        codec.unpack(b"XXXX" + blob[4:])
    with pytest.raises(ValueError):
        codec.unpack(blob[:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,R", [("clc", 1), ("tcm", 0)])
def test_engine_matches_model_codec(dev, kind, R):
    from clc_amd import codec
    from clc_amd import models as pm
    from clc_amd.recipe import apply_weight_recipe, synthetic_image

    m = pm.CLC(N=64, num_ref_frames=R) if kind == "clc" else pm.TCM(N=64)
    apply_weight_recipe(m, 0)
    m = m.to(dev).eval()
    m.update(force=True)
    B = 3
    x = torch.cat([synthetic_image(1, 256, 256, 100 + 7 * i, smooth=True) for i in range(B)]).to(dev)
    refs = [torch.cat([synthetic_image(1, 256, 256, 200 + 7 * i + j, smooth=True) for i in range(B)]).to(dev) for j in range(R)]
    want = []
    for i in range(B):
        ri = [r[i:i + 1] for r in refs]
        # (the eager launch-by-launch methods: model.compress() itself rides on a CodecEngine for single images, test below)
        enc = m._compress_eager(x[i:i + 1], ri if R else None)
        dec = m._decompress_eager(enc["strings"], enc["shape"], ri if R else None)
        want.append((enc, dec["x_hat"]))
    results = {}
    for use_graph in (True, False):
        eng = codec.CodecEngine(m, threads=4, use_graph=use_graph)
        outs = eng.compress(x, refs)
        assert len(outs) == B
        for i, (o, (enc, _)) in enumerate(zip(outs, want)):
            assert o["strings"][0][0] == enc["strings"][0][0], f"image {i}: y stream differs from model.compress (graph={use_graph})"
            assert o["strings"][1][0] == enc["strings"][1][0], f"image {i}: z stream differs"
            assert tuple(o["shape"]) == tuple(enc["shape"])
        x_hat = eng.decompress(outs, refs)
        for i in range(B):
            assert torch.equal(x_hat[i:i + 1], want[i][1]), f"image {i}: decoded image differs from model.decompress (graph={use_graph})"
        # a second batch through the SAME captured graphs (other content, other order)
        perm = [2, 0, 1]
        outs2 = eng.compress(x[perm], [r[perm] for r in refs])
        for k, i in enumerate(perm):
            assert outs2[k]["strings"] == outs[i]["strings"], "an image's streams depend on its position in the batch"
        x_hat2 = eng.decompress(outs2, [r[perm] for r in refs])
        assert torch.equal(x_hat2, x_hat[perm])
        # through the container
        blobs = [codec.pack(o["strings"], o["shape"], (256, 256), n_refs=R) for o in outs]
        items = [codec.unpack_item(b) for b in blobs]      # (the header's tag travels with the item and is checked by decompress)
        assert torch.equal(eng.decompress(items, refs), x_hat)
        results[use_graph] = (outs, x_hat)
    assert torch.equal(results[True][1], results[False][1])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,R,H,W", [("clc", 1, 256, 256), ("clc", 3, 256, 384), ("tcm", 0, 256, 256)])
def test_reference_surface_single_image_codec_rides_on_captured_graphs(dev, kind, R, H, W):
    """model.compress() / decompress() — what /root/reference/eval_CLC.py:314-338 calls, one padded image at a time
    (CLC_run.py:629-716, 738-814) — run on the captured segments of a lazily built CodecEngine: byte-identical streams and bit-identical
    images to the eager methods; a second image of the same shape reuses the graphs; re-homed parameters (what TrainEngine does) or a
    changed precision mode rebuild them; batches > 1 and CLC_CODEC_GRAPH=0 keep the eager path."""
    import clc_amd
    from clc_amd import models as pm
    from clc_amd.models import clc as mclc
    from clc_amd.recipe import apply_weight_recipe, synthetic_image

    m = pm.CLC(N=64, num_ref_frames=R) if kind == "clc" else pm.TCM(N=64)
    apply_weight_recipe(m, 0)
    m = m.to(dev).eval()
    m.update(force=True)
    imgs = [synthetic_image(1, H, W, 500 + 3 * i, smooth=True).to(dev) for i in range(2)]
    refs = [[synthetic_image(1, H, W, 600 + 10 * i + j, smooth=True).to(dev) for j in range(R)] for i in range(2)]
    engines = []
    engines_first_strings = None
    for x, rf in zip(imgs, refs):
        a = m.compress(x, rf) if R else m.compress(x)
        if engines_first_strings is None:
            engines_first_strings = a["strings"]
        b = m._compress_eager(x, rf if R else None)
        assert a["strings"] == b["strings"] and tuple(a["shape"]) == tuple(b["shape"])
        da = m.decompress(a["strings"], a["shape"], rf) if R else m.decompress(a["strings"], a["shape"])
        db = m._decompress_eager(b["strings"], b["shape"], rf if R else None)
        assert torch.equal(da["x_hat"], db["x_hat"])
        engines.append(m.__dict__.get("_codec_eng"))
    eng = engines[0]
    assert eng is not None and engines[1] is eng and len(eng._enc) == 1 and len(eng._dec) == 1     # one set of graphs, reused
    assert next(iter(eng._enc.values())).graph is not None
    assert not m.training
    # weights changed IN PLACE through `.data` (no version counter moves): the captured graphs read the live storage and compute every derived
    # image (fragment-order filters of the halo kernel, GDN re-parametrisation) inside the graph -> same engine, new streams, still == eager
    with torch.no_grad():
        for q in m.parameters():
            if q.dim() == 4:
                q.data.mul_(1.003)
    a1 = m.compress(imgs[0], refs[0]) if R else m.compress(imgs[0])
    b1 = m._compress_eager(imgs[0], refs[0] if R else None)
    assert m.__dict__["_codec_eng"] is eng and a1["strings"] == b1["strings"] and a1["strings"] != engines_first_strings
    d1 = m.decompress(a1["strings"], a1["shape"], refs[0]) if R else m.decompress(a1["strings"], a1["shape"])
    assert torch.equal(d1["x_hat"], m._decompress_eager(b1["strings"], b1["shape"], refs[0] if R else None)["x_hat"])
    # re-homed parameters: the captured graphs would read the old storage -> the engine must be rebuilt, results unchanged
    with torch.no_grad():
        for q in m.parameters():
            q.data = q.data.clone(memory_format=torch.preserve_format)
    a2 = m.compress(imgs[0], refs[0]) if R else m.compress(imgs[0])
    assert m.__dict__["_codec_eng"] is not eng and a2["strings"] == (m._compress_eager(imgs[0], refs[0] if R else None))["strings"]
    # batch > 1: the reference's joint y stream (eager path), untouched
    xb = torch.cat(imgs)
    rb = [torch.cat([refs[0][j], refs[1][j]]) for j in range(R)]
    eb = m.compress(xb, rb) if R else m.compress(xb)
    assert len(eb["strings"][0]) == 1 and len(eb["strings"][1]) == 2
    # the model's own engine keeps a bounded number of captured sizes (least recently used out first)
    eng2 = m.__dict__["_codec_eng"]
    assert eng2.max_plans == mclc.CODEC_GRAPH_PLANS
    eng2.max_plans = 1
    other = torch.nn.functional.pad(imgs[0], (0, 128, 0, 0))      # another signature
    ro = [torch.nn.functional.pad(r, (0, 128, 0, 0)) for r in refs[0]]
    ao = m.compress(other, ro) if R else m.compress(other)
    assert len(eng2._enc) == 1 and ao["strings"] == m._compress_eager(other, ro if R else None)["strings"]
    eng2.max_plans = mclc.CODEC_GRAPH_PLANS
    old = mclc.CODEC_GRAPH
    mclc.CODEC_GRAPH = False
    try:
        assert m._codec_engine(imgs[0]) is None
        a3 = m.compress(imgs[0], refs[0]) if R else m.compress(imgs[0])
        assert a3["strings"] == a2["strings"]
    finally:
        mclc.CODEC_GRAPH = old
