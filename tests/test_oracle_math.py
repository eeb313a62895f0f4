"""CPU tests of the oracle's building blocks (the checker must itself be checked):
expf vs this container's libm, binary16 conversions vs numpy, the SH basis vs float64
formulas, camera pose math, and basic invariants of the oracle renderer."""
import ctypes as C
import ctypes.util
import struct
from decimal import Decimal, getcontext

import numpy as np

import cases


def test_exp2f_table_rederived_from_first_principles(orc):
    """tab[i] = asuint64(2^(i/32) correctly rounded) - (i << 47): regenerate and compare with what
    the oracle uses by probing orc_expf at x = ln2 * i/32 (r == 0 path) -- indirectly -- and check
    the constants literally against the header text of both implementations."""
    getcontext().prec = 60
    want = []
    for i in range(32):
        d = float(Decimal(2) ** (Decimal(i) / Decimal(32)))
        bits = struct.unpack("<Q", struct.pack("<d", d))[0]
        want.append((bits - (i << 47)) & 0xFFFFFFFFFFFFFFFF)
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rel in ("oracle/mnv_oracle.c", "mega-nerf-viewer_amd/csrc/mnv_device.h"):
        text = open(os.path.join(root, rel)).read()
        for v in want:
            assert "0x%016xULL" % v in text, (rel, hex(v))


def test_expf_matches_libm_bit_for_bit(orc):
    """glibc's expf and the oracle's restatement agree on a dense strided sample of all binary32
    values (an exhaustive run over all 2^32 inputs in this container found 2 mismatches, both 1 ulp,
    from the FMA-contracted ifunc variant glibc selects on this CPU)."""
    libm = C.CDLL(ctypes.util.find_library("m"))
    libm.expf.restype = C.c_float
    libm.expf.argtypes = [C.c_float]
    rng = np.random.default_rng(0)
    xs = np.concatenate([
        np.arange(0, 2**32, 2**32 // (1 << 16), dtype=np.uint64).astype(np.uint32).view(np.float32),
        rng.uniform(-110, 90, 60000).astype(np.float32),
        np.float32([0.0, -0.0, 1.0, -1.0, 88.0, 88.7228, 88.73, -87.3, -103.0, -103.9, -104.0, -1e9, 1e9, np.inf, -np.inf]),
    ])
    bad = 0
    for x in xs:
        a, b = orc.lib().orc_expf(float(x)), libm.expf(float(x))
        if np.isnan(a) and np.isnan(b):
            continue
        if np.float32(a).view(np.uint32) != np.float32(b).view(np.uint32):
            bad += 1
            assert abs(int(np.float32(a).view(np.int32)) - int(np.float32(b).view(np.int32))) <= 1
    assert bad <= 2


def test_half_conversions_match_numpy(orc):
    h = np.arange(65536, dtype=np.uint16)
    want = h.view(np.float16).astype(np.float32)
    got = np.array([orc.lib().orc_half_to_float(int(v)) for v in h], np.float32)
    m = ~np.isnan(want)
    assert np.array_equal(got[m].view(np.uint32), want[m].view(np.uint32)) and np.isnan(got[~m]).all()
    rng = np.random.default_rng(1)
    f = np.concatenate([rng.normal(0, 10, 20000), rng.uniform(-7e4, 7e4, 5000), rng.normal(0, 1e-5, 5000),
                        [0.0, -0.0, 65504.0, 65519.9, 65520.0, 6.1e-5, 5.96e-8, 2.98e-8, 2.99e-8]]).astype(np.float32)
    want16 = f.astype(np.float16).view(np.uint16)
    got16 = np.array([orc.lib().orc_float_to_half(float(v)) for v in f], np.uint16)
    assert np.array_equal(got16, want16)


def test_sh_basis_close_to_float64_formulas(orc):
    rng = np.random.default_rng(2)
    for _ in range(50):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        x, y, z = d
        ref = np.zeros(25)
        ref[0] = 0.28209479177387814
        ref[1], ref[2], ref[3] = -0.4886025119029199 * y, 0.4886025119029199 * z, -0.4886025119029199 * x
        ref[4], ref[5] = 1.0925484305920792 * x * y, -1.0925484305920792 * y * z
        ref[6] = 0.31539156525252005 * (2 * z * z - x * x - y * y)
        ref[7], ref[8] = -1.0925484305920792 * x * z, 0.5462742152960396 * (x * x - y * y)
        ref[9] = -0.5900435899266435 * y * (3 * x * x - y * y)
        ref[10] = 2.890611442640554 * x * y * z
        ref[11] = -0.4570457994644658 * y * (4 * z * z - x * x - y * y)
        ref[12] = 0.3731763325901154 * z * (2 * z * z - 3 * x * x - 3 * y * y)
        ref[13] = -0.4570457994644658 * x * (4 * z * z - x * x - y * y)
        ref[14] = 1.445305721320277 * z * (x * x - y * y)
        ref[15] = -0.5900435899266435 * x * (x * x - 3 * y * y)
        ref[16] = 2.5033429417967046 * x * y * (x * x - y * y)
        ref[17] = -1.7701307697799304 * y * z * (3 * x * x - y * y)
        ref[18] = 0.9461746957575601 * x * y * (7 * z * z - 1)
        ref[19] = -0.6690465435572892 * y * z * (7 * z * z - 3)
        ref[20] = 0.10578554691520431 * (z * z * (35 * z * z - 30) + 3)
        ref[21] = -0.6690465435572892 * x * z * (7 * z * z - 3)
        ref[22] = 0.47308734787878004 * (x * x - y * y) * (7 * z * z - 1)
        ref[23] = -1.7701307697799304 * x * z * (x * x - 3 * y * y)
        ref[24] = 0.6258357354491761 * (x * x * (x * x - 3 * y * y) - y * y * (3 * x * x - y * y))
        for nb in (1, 4, 9, 16, 25):
            out = (C.c_float * 25)()
            orc.lib().orc_sh_basis(nb, (C.c_float * 3)(*d.astype(np.float32)), out)
            got = np.array(list(out))
            n_set = nb if nb > 1 else 1
            assert np.allclose(got[:n_set], ref[:n_set], atol=3e-6)
            assert np.all(got[n_set:] == 0)


def test_camera_pose_matches_numpy_float32(orc):
    f32 = np.float32
    center, back, up = f32([-3.55, 0.0, 3.55]), f32([-0.7071068, 0.0, 0.7071068]), f32([0, 0, 1])
    out = (C.c_float * 12)()
    arr = lambda v: (C.c_float * 3)(*v)
    orc.lib().orc_camera_pose(arr(center), arr(back), arr(up), out)
    got = np.array(list(out), f32)

    def norm(v):
        d = f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))
        return v * f32(f32(1) / np.sqrt(d))

    def cross(a, b):
        return f32([f32(a[1] * b[2]) - f32(b[1] * a[2]), f32(a[2] * b[0]) - f32(b[2] * a[0]), f32(a[0] * b[1]) - f32(b[0] * a[1])])

    b = norm(back)
    r = norm(cross(up, b))
    u = cross(b, r)
    want = np.concatenate([r, u, b, center]).astype(f32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_oracle_render_invariants(mnv, orc):
    spec = cases.CASES["sh4_d6"]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    ot = orc.tree_from_view(tree.host_view())
    full = orc.render(ot, cam.c, opt, want_rgba8=True, want_steps=True)
    one = orc.render(ot, cam.c, opt, n_threads=1)
    assert np.array_equal(cases.bits(full["rgba"]), cases.bits(one["rgba"]))           # thread-count invariant
    assert full["counters"].as_dict() == one["counters"].as_dict()
    tile = (17, 9, 100, 77)
    part = orc.render(ot, cam.c, opt, tile=tile)["rgba"]
    assert np.array_equal(cases.bits(part), cases.bits(full["rgba"][9:86, 17:117]))       # tile invariant
    a = full["rgba"][..., 3]
    assert a.min() >= 0 and a.max() <= 1 and np.isfinite(full["rgba"]).all()
    c = full["counters"].as_dict()
    assert c["rays"] == cam.width * cam.height and c["steps"] == int(full["steps"].sum())
    assert c["levels"] >= c["steps"] >= c["hits"] and c["max_steps"] == int(full["steps"].max())
    # SURVEY 8(d) byte formula
    assert orc.algorithmic_bytes(full["counters"], 1, 4) == 16 * c["rays"] + 4 * c["levels"] + 2 * c["steps"] + 24 * c["hits"]
    # u8 pack = truncation of the float output (renderer_kernel.cu:237)
    assert np.array_equal(full["rgba8"][..., :3], np.clip(full["rgba"][..., :3] * np.float32(255), 0, 255).astype(np.uint8))
    assert np.all(full["rgba8"][..., 3] == 255)


def test_oracle_ray_miss_and_depth_mode(mnv, orc):
    spec = cases.CASES["ray_miss"]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    ot = orc.tree_from_view(tree.host_view())
    r = orc.render(ot, cam.c, opt)
    assert r["counters"].rays_in_bbox == 0
    assert np.all(r["rgba"][..., :3] == opt.background_brightness) and np.all(r["rgba"][..., 3] == 0)
    opt.render_depth = True  # a miss in depth mode sets alpha = 1 (rt_core.cuh:196)
    r = orc.render(ot, cam.c, opt)
    assert np.all(r["rgba"][..., 3] == 1) and np.all(r["rgba"][..., :3] == 0)


def test_mlp_restatement_against_numpy(mnv, orc):
    """oracle/mnv_oracle.c:orc_mlp_forward (the build's own network, parity unpinned) against an independent
    float64 numpy evaluation of the same definition (binary16 weights / activations)."""
    import mlp_cases
    desc = mnv.mlp_desc(n_clusters=3, pos_octaves=3, dir_octaves=2, need_viewdir=True, n_embeddings=5, embedding_dim=4,
                        hidden_width=64, hidden_layers=2, out_dim=7, center=(0.1, 0.2, -0.3), inv_extent=(0.5, 0.4, 0.8))
    params = mlp_cases.make_params(mnv, desc, seed=9)
    x, cluster = mlp_cases.make_samples(desc, 300, seed=10)
    got = orc.mlp_forward(desc, params, cluster, x)
    per = mnv.Mlp.param_count(desc)

    def tri(t):
        return 4.0 * np.abs(t - np.floor(t + 0.5)) - 1.0

    def block(v, octaves):
        out = [v]
        for k in range(octaves):
            out += [tri(v * 2.0 ** k), tri(v * 2.0 ** k + 0.25)]
        return np.concatenate(out)

    f16 = lambda a: np.asarray(a, np.float64).astype(np.float16).astype(np.float64)  # noqa: E731
    for row in range(x.shape[0]):
        c = int(cluster[row])
        if c < 0 or c >= desc.n_clusters:
            assert np.all(got[row] == 0)
            continue
        P = params[c * per:(c + 1) * per].astype(np.float64)
        p = (x[row, :3] - np.float32(list(desc.center))).astype(np.float32) * np.float32(list(desc.inv_extent))
        enc = np.concatenate([block(p.astype(np.float64), 3), block(x[row, 3:6].astype(np.float64), 2),
                              P[per - 20:].reshape(5, 4)[int(x[row, 6])]])
        h, off, dims = f16(enc), 0, [(64, enc.size), (64, 64), (7, 64)]
        for li, (o, i) in enumerate(dims):
            w, b = P[off:off + o * i].reshape(o, i), P[off + o * i: off + o * i + o]
            off += o * i + o
            h = w @ h + b
            if li < 2:
                h = f16(np.maximum(h, 0))
        assert np.allclose(got[row], h, rtol=2e-3, atol=2e-3)


def test_oracle_onscreen_inputs_reduce_to_the_offscreen_call(mnv, orc):
    """orc_render_voxels_ex (the reference's offscreen == false call shape, renderer_kernel.cu:230-234,277-280): a depth image of 1e9f
    everywhere is the offscreen t_max, a white image under the volume is background_brightness 1, and a depth image of zeros leaves the
    image as it was."""
    import cases

    spec = cases.CASES["sh4_d6"]
    tree = cases.make_tree(mnv, spec["tree"])
    cam, opt = cases.make_camera(mnv, spec["camera"]), cases.make_options(mnv, spec["options"])
    t = orc.tree_from_view(tree.host_view())
    h, w = cam.height, cam.width
    plain = orc.render(t, cam.c, opt, want_rgba8=True)
    far = orc.render(t, cam.c, opt, want_rgba8=True, tmax_px=np.full((h, w), 1e9, np.float32))
    assert np.array_equal(plain["rgba"].view(np.uint32), far["rgba"].view(np.uint32)) and np.array_equal(plain["rgba8"], far["rgba8"])
    opt.background_brightness = 1.0
    white = np.full((h, w, 4), 255, np.uint8)
    a = orc.render(t, cam.c, opt)["rgba"]
    b = orc.render(t, cam.c, opt, rgba8_init=white)["rgba"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    rng = np.random.default_rng(3)
    image = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    shut = orc.render(t, cam.c, opt, want_rgba8=True, tmax_px=np.zeros((h, w), np.float32), rgba8_init=image)
    assert np.all(shut["rgba"][..., 3] == 0.0) and np.array_equal(shut["rgba8"][..., :3], image[..., :3]) and np.all(shut["rgba8"][..., 3] == 255)
