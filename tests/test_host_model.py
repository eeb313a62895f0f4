"""Host-side data model behind the C ABI: DataFormat, Camera, N3Tree (.npz loader/writer incl. deflate,
scalar invradius, VQ decode), deterministic synthetic trees."""
import hashlib
import os

import numpy as np
import pytest

import cases


@pytest.mark.parametrize("s,fmt,basis", [("SH9", 1, 9), ("SH16", 1, 16), ("RGBA", 0, -1), ("SH", 0, -1), ("RGBA3", 0, 3), ("XYZ4", 0, 4), ("", 0, -1), ("SH25", 1, 25)])
def test_data_format_parse(mnv, s, fmt, basis):
    # reference src/data_format.cpp:5-24: leading alphabetic run names the format, atoi of the rest
    assert mnv.parse_data_format(s) == (fmt, basis)


def test_data_format_to_string(mnv):
    assert mnv.data_format_to_string(1, 9) == "SH9" and mnv.data_format_to_string(0, -1) == "RGBA"


def test_camera_defaults_match_reference_ctor(mnv, orc):
    cam = mnv.Camera()  # 256x256, fx 1111, fy = fx, cx = cy = 128 (camera.cpp:29-46)
    c = cam.c
    assert (c.width, c.height, c.fx, c.fy, c.cx, c.cy) == (256, 256, 1111.0, 1111.0, 128.0, 128.0)
    import ctypes as C
    out = (C.c_float * 12)()
    f3 = lambda v: (C.c_float * 3)(*v)
    orc.lib().orc_camera_pose(f3([-3.55, 0.0, 3.55]), f3([-0.7071068, 0.0, 0.7071068]), f3([0, 0, 1]), out)
    assert np.array_equal(cam.c2w.view(np.uint32), np.array(list(out), np.float32).view(np.uint32))
    odd = mnv.Camera(101, 77, 500.0, 400.0)  # integer halving of the principal point
    assert (odd.c.cx, odd.c.cy, odd.c.fy) == (50.0, 38.0, 400.0)


def _svox_arrays(rng, cap=5, data_dim=28):
    child = np.zeros((cap, 2, 2, 2), np.int32)
    child[0, 0, 0, 1], child[0, 1, 1, 0], child[1, 0, 1, 0], child[2, 1, 0, 0] = 1, 2, 2, 2
    parent_depth = np.array([[-1, 0], [1, 1], [6, 1], [10, 2], [20, 2]], np.int32)
    data = rng.normal(size=(cap, 2, 2, 2, data_dim)).astype(np.float16)
    return child, parent_depth, data


@pytest.mark.parametrize("compressed", [False, True])
@pytest.mark.parametrize("scalar_radius", [False, True])
def test_open_numpy_written_svox_npz(mnv, tmp_path, compressed, scalar_radius):
    """Keys / dtypes as read by reference n3tree.cpp:28-107,177-188."""
    rng = np.random.default_rng(0)
    child, parent_depth, data = _svox_arrays(rng)
    kw = dict(data_dim=np.int64(28), data_format=np.array("SH9"), offset=np.float32([0.5, 0.4, 0.3]), child=child,
              parent_depth=parent_depth, data=data)
    if scalar_radius:
        kw["invradius"] = np.float64(0.37)
    else:
        kw["invradius3"] = np.float32([0.5, 0.25, 0.125])
    path = str(tmp_path / "t.npz")
    (np.savez_compressed if compressed else np.savez)(path, **kw)
    t = mnv.N3Tree.open(path)
    v = t.host_view()
    assert (v.N, v.data_dim, v.format, v.basis_dim, v.capacity) == (2, 28, 1, 9, 5)
    assert list(v.offset) == [np.float32(0.5), np.float32(0.4), np.float32(0.3)]
    want_scale = [np.float32(0.37)] * 3 if scalar_radius else [0.5, 0.25, 0.125]
    assert list(v.scale) == want_scale
    d, c, p = t.host_arrays()
    assert np.array_equal(d, data.view(np.uint16).reshape(5, 8, 28))
    assert np.array_equal(c, child.reshape(5, 8)) and np.array_equal(p, parent_depth[:, 0])


def test_save_npz_round_trip_and_numpy_readable(mnv, tmp_path):
    t = cases.make_tree(mnv, cases.CASES["sh9_d7_aniso"]["tree"])
    path = str(tmp_path / "rt.npz")
    t.save_npz(path)
    z = np.load(path)
    d, c, p = t.host_arrays()
    cap = t.capacity
    assert int(z["data_dim"]) == 28 and str(z["data_format"]) == "SH9"
    assert z["child"].shape == (cap, 2, 2, 2) and z["data"].shape == (cap, 2, 2, 2, 28) and z["data"].dtype == np.float16
    assert np.array_equal(z["child"].reshape(cap, 8), c) and np.array_equal(z["data"].view(np.uint16).reshape(cap, 8, 28), d)
    assert np.array_equal(z["parent_depth"][:, 0], p) and np.array_equal(z["invradius3"], np.float32([0.5, 0.25, 0.125]))
    t2 = mnv.N3Tree.open(path)
    d2, c2, p2 = t2.host_arrays()
    assert np.array_equal(d, d2) and np.array_equal(c, c2) and np.array_equal(p, p2)


def test_missing_file_gives_empty_tree_and_bad_files_raise(mnv, tmp_path):
    t = mnv.N3Tree.open(str(tmp_path / "nope.npz"))   # printf + N == 0, n3tree.cpp:19-22
    assert t.host_view().N == 0
    rng = np.random.default_rng(0)
    child, parent_depth, data = _svox_arrays(rng)
    base = dict(data_dim=np.int64(28), data_format=np.array("SH9"), offset=np.float32([0, 0, 0]), invradius3=np.float32([1, 1, 1]),
                child=child, parent_depth=parent_depth, data=data)
    bad = dict(base, data=data.astype(np.float32))                      # "data must be stored in half precision"
    np.savez(str(tmp_path / "a.npz"), **bad)
    with pytest.raises(mnv.MnvError):
        mnv.N3Tree.open(str(tmp_path / "a.npz"))
    bad = dict(base, parent_depth=parent_depth[:4])                      # "data and parent sizes not aligned"
    np.savez(str(tmp_path / "b.npz"), **bad)
    with pytest.raises(mnv.MnvError):
        mnv.N3Tree.open(str(tmp_path / "b.npz"))
    (tmp_path / "c.npz").write_bytes(b"not a zip file at all, definitely")
    with pytest.raises(mnv.MnvError):
        mnv.N3Tree.open(str(tmp_path / "c.npz"))
    with pytest.raises(mnv.MnvError):
        mnv.N3Tree.open(str(tmp_path / "c.txt"))


def test_vq_npz_decodes_to_channel_major_rows(mnv, tmp_path):
    """VQ PlenOctree files (quant_colors / quant_map / data_retained / sigma).  The reference's decode
    loop is defective; this pins the build's documented layout instead
    (parity with the reference loader: UNPINNED)."""
    rng = np.random.default_rng(3)
    cap, n_basis, n_retain = 3, 4, 1
    n_q = n_basis - n_retain
    child = np.zeros((cap, 2, 2, 2), np.int32)
    child[0, 0, 0, 0], child[0, 1, 1, 1] = 1, 2
    parent_depth = np.array([[-1, 0], [0, 1], [7, 1]], np.int32)
    book = rng.normal(size=(n_q, 65536, 3)).astype(np.float16)
    qmap = rng.integers(0, 65536, size=(n_q, cap, 2, 2, 2)).astype(np.uint16)
    retained = rng.normal(size=(n_retain, cap, 2, 2, 2, 3)).astype(np.float16)
    sigma = rng.uniform(0, 50, size=(cap, 2, 2, 2)).astype(np.float16)
    path = str(tmp_path / "vq.npz")
    np.savez(path, data_dim=np.int64(13), data_format=np.array("SH4"), offset=np.float32([0.5] * 3), invradius3=np.float32([0.5] * 3),
             child=child, parent_depth=parent_depth, quant_colors=book, quant_map=qmap, data_retained=retained, sigma=sigma)
    t = mnv.N3Tree.open(path)
    d, _, _ = t.host_arrays()
    d = d.view(np.float16).reshape(cap, 8, 13)
    want = np.zeros((cap, 8, 13), np.float16)
    for ch in range(3):
        want[:, :, ch * n_basis] = retained[0, ..., ch].reshape(cap, 8)
        for b in range(n_q):
            want[:, :, ch * n_basis + n_retain + b] = book[b, qmap[b].reshape(cap, 8), ch]
    want[:, :, 12] = sigma.reshape(cap, 8)
    assert np.array_equal(d.view(np.uint16), want.view(np.uint16))


# sha256 of (child, data) of two generator configurations: pins the deterministic generator so that
# fixtures made in one container match trees rebuilt on the GPU box
_SYNTH_PINS = {
    "cfg1_sh1_d4": None,
    "shell_d7_sh9": None,
}


def _tree_digest(t):
    d, c, p = t.host_arrays()
    h = hashlib.sha256()
    for a in (c, d, p):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_synth_trees_are_deterministic_and_well_formed(mnv):
    golden = os.path.join(os.path.dirname(__file__), "golden", "synth_digests.txt")
    pins = dict(l.split() for l in open(golden).read().splitlines() if l.strip())
    for name in _SYNTH_PINS:
        spec = cases.CASES[name]["tree"]
        t1, t2 = cases.make_tree(mnv, spec), cases.make_tree(mnv, spec)
        assert _tree_digest(t1) == _tree_digest(t2) == pins[name], name
        d, c, p = t1.host_arrays()
        cap = t1.capacity
        # every non-root chunk is referenced exactly once, by the voxel its parent entry names
        tgt = (np.arange(cap)[:, None] + c)[c != 0]
        assert np.array_equal(np.sort(tgt), np.arange(1, cap))
        src = np.argwhere(c != 0)
        assert np.array_equal(p[tgt], src[:, 0] * 8 + src[:, 1]) and p[0] == -1


def test_cfg2_tree_statistics(mnv):
    """BASELINE.json configs[1]: the depth-10 SH9 shell has 1,499,569 chunks."""
    t = cases.make_tree(mnv, cases.CFG2_TREE)
    v = t.host_view()
    assert (v.capacity, v.data_dim, v.basis_dim, v.format) == (1499569, 28, 9, 1)
    d, c, _ = t.host_arrays()
    assert int((c != 0).sum()) == v.capacity - 1
    sig = d[:, :, 27].view(np.float16)
    dense = sig > 0
    assert int(dense.sum()) > 4_000_000 and float(sig[dense].min()) >= 50 and float(sig.max()) <= 400


def test_npz_reader_rejects_malformed_files_without_hanging(mnv, tmp_path):
    """Fuzzed copies of a valid tree file (truncations, byte flips, a shape tuple without digits, sizes larger than the file)
    either load or raise MnvError -- the reader validates every size it takes from the file (host/npz.cpp).  The same inputs
    ran clean under AddressSanitizer / UBSan on the host build of the reader (3000 cases)."""
    import cases
    tree = cases.make_tree(mnv, cases.CASES["cfg1_sh1_d4"]["tree"])
    good = tmp_path / "good.npz"
    tree.save_npz(str(good))
    buf = good.read_bytes()
    rng = np.random.default_rng(3)
    bad = tmp_path / "bad.npz"
    loaded = rejected = 0
    for t in range(150):
        b = bytearray(buf)
        if t % 3 == 0:
            b = b[: int(rng.integers(0, len(b)))]
        elif t % 3 == 1:
            for _ in range(8):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        else:  # corrupt a size / offset field of the end-of-central-directory record or of a central-directory entry
            pos = bytes(b).rfind(b"PK\x05\x06") if t % 2 else bytes(b).find(b"PK\x01\x02")
            off = pos + int(rng.choice([12, 16, 20, 24, 42]))
            b[off:off + 4] = int(rng.integers(0, 2 ** 32)).to_bytes(4, "little")
        bad.write_bytes(bytes(b))
        try:
            mnv.N3Tree.open(str(bad))
            loaded += 1
        except mnv.MnvError:
            rejected += 1
    assert loaded + rejected == 150 and rejected > 50
    # a header whose shape tuple holds no digits used to spin forever
    i = buf.find(b"'shape': (")
    b = bytearray(buf)
    b[i + 10:i + 12] = b"a,"
    bad.write_bytes(bytes(b))
    with pytest.raises(mnv.MnvError):
        mnv.N3Tree.open(str(bad))


def test_camera_drag_matches_reference_camera(mnv):
    """Camera::begin_drag / drag_update / end_drag (include/camera.hpp:22-25, src/camera.cpp:132-187): 64 drags -- rotations about the camera
    and about the origin, pans, drags that wrap the azimuth or run into the pole guard -- against the poses the reference's own Camera (glm)
    produced (tests/golden/make_camera_drag_goldens.py, run inside oracle/_ref on the GPU box).  Kept for API compatibility; the window
    that used them is out of scope."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_camera_drag.npz"))
    kinds = set()
    n_exact = 0
    for d, want in zip(g["inputs"], g["outputs"]):
        cam = mnv.Camera(int(d[0]), int(d[1]), float(np.float32(d[2])))
        c, b, o, m = mnv.camera_drag(cam, np.float32(d[3:6]), np.float32(d[6:9]), np.float32(d[9:12]), np.float32(d[12:15]), np.float32(d[15]), bool(d[16]),
                                     bool(d[17]), (np.float32(d[18]), np.float32(d[19])), (np.float32(d[20]), np.float32(d[21])))
        got = np.concatenate([c, b, o, m])
        assert np.allclose(got, want, rtol=0, atol=2e-6), (d, got, want)
        n_exact += int(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
        kinds.add((bool(d[16]), bool(d[17])))
    assert kinds == {(False, False), (True, False), (False, True)} and n_exact >= 48   # (cosf / sinf of glibc on both sides: mostly bit-identical)


def test_camera_drag_properties(mnv):
    """What the drag does, without the goldens: a pan slides center (and origin with about_origin) along the start's right / up vectors by
    -2 * speed / max(w, h) per pixel; a rotation keeps |v_back| = 1 and, about the origin, the distance centre-origin; a tilt over the pole
    is refused; a drag of zero pixels changes nothing."""
    cam = mnv.Camera(800, 600, 700.0)
    c0, b0, up, org = np.float32([-3.55, 0.0, 3.55]), np.float32([-0.7071068, 0.0, 0.7071068]), (0.0, 0.0, 1.0), np.float32([0.3, -0.2, 0.1])
    c, b, o, m = mnv.camera_drag(cam, c0, b0, up, org, 1.0, False, False, (10, 10), (10, 10))
    assert np.allclose(c, c0) and np.allclose(b, b0, atol=1e-7) and np.allclose(o, org)
    c, b, o, m = mnv.camera_drag(cam, c0, b0, up, org, 2.0, True, True, (100, 100), (300, 180))
    right, upv = np.float32([0.0, -1.0, 0.0]), np.float32([0.7071068, 0.0, 0.7071068])
    k = -2.0 * 2.0 / 800
    assert np.allclose(c, c0 + 200 * k * right - 80 * k * upv, atol=1e-5) and np.allclose(o - org, c - c0, atol=1e-6)
    c, b, o, m = mnv.camera_drag(cam, c0, b0, up, org, 1.0, False, True, (100, 100), (260, 40))
    assert abs(np.linalg.norm(b) - 1) < 1e-6 and abs(np.linalg.norm(c - org) - np.linalg.norm(c0 - org)) < 1e-5 and not np.allclose(c, c0)
    c, b, o, m = mnv.camera_drag(cam, c0, b0, up, org, 1.0, False, False, (100, 100), (100, 100 + 1000))   # 2.5 rad of tilt from 45 degrees: over the pole
    assert np.allclose(b, b0, atol=1e-7)
