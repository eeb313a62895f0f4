"""Deterministic parameters and inputs for the build's own sub-module MLP (tests, smoke, bench)."""
import numpy as np


def make_params(mnv, desc, seed=0, gain=1.6):
    """He-style random binary16 parameters in the blob order of include/mnv.h (cluster-major)."""
    rng = np.random.default_rng(seed)
    n_pos = 3 + 6 * desc.pos_octaves
    n_dir = (3 + 6 * desc.dir_octaves) if desc.need_viewdir else 0
    emb = desc.embedding_dim if desc.n_embeddings > 0 else 0
    in_dim, w = n_pos + n_dir + emb, desc.hidden_width
    blobs = []
    for _ in range(desc.n_clusters):
        dims = [(w, in_dim)] + [(w, w)] * (desc.hidden_layers - 1) + [(desc.out_dim, w)]
        for o, i in dims:
            blobs.append((rng.standard_normal((o, i)) * gain / np.sqrt(i)).astype(np.float16).reshape(-1))
            blobs.append((rng.standard_normal(o) * 0.1).astype(np.float16))
        if emb:
            blobs.append((rng.standard_normal((desc.n_embeddings, emb)) * 0.5).astype(np.float16).reshape(-1))
    params = np.concatenate(blobs)
    assert params.size == desc.n_clusters * mnv.Mlp.param_count(desc)
    return params


def make_samples(desc, n, seed=1, invalid_frac=0.05):
    rng = np.random.default_rng(seed)
    cols = 3 + (3 if desc.need_viewdir else 0) + (1 if desc.n_embeddings > 0 else 0)
    x = np.zeros((n, cols), np.float32)
    x[:, :3] = rng.uniform(-1.5, 1.5, (n, 3))
    if desc.need_viewdir:
        d = rng.standard_normal((n, 3))
        x[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    if desc.n_embeddings > 0:
        x[:, -1] = rng.integers(0, desc.n_embeddings, n)
    cluster = rng.integers(0, desc.n_clusters, n).astype(np.int16)
    bad = rng.random(n) < invalid_frac
    cluster[bad] = np.where(rng.random(bad.sum()) < 0.5, -1, desc.n_clusters + 3)
    return x, cluster
