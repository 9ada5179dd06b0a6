"""CPU: known-answer and gradient tests of the oracle's tiny-cuda-nn restatement (parity unpinned: no reference
vectors exist for these) -- they pin the restatement to hand-checkable facts."""
import numpy as np
import torch

from oracle import tcnn_ref as tr


def _meta():
    return tr.grid_meta(16, 592)


def test_level0_dense_index_formula():
    m = _meta()
    lvl = m.levels[0]
    g = torch.tensor([[3, 5, 7]])
    idx = tr.grid_index(g[:, 0], g[:, 1], g[:, 2], lvl)
    assert int(idx) == 3 + 16 * 5 + 256 * 7            # x + 16 y + 256 z  


def test_hash_primes_and_wrap():
    m = _meta()
    lvl = m.levels[10]
    assert lvl.hashed
    gx, gy, gz = torch.tensor([100]), torch.tensor([200]), torch.tensor([300])
    want = (100 ^ ((200 * 2654435761) & 0xFFFFFFFF) ^ ((300 * 805459861) & 0xFFFFFFFF)) % lvl.size
    assert int(tr.grid_index(gx, gy, gz, lvl)) == want


def test_vertex_hit_returns_table_row():
    m = _meta()
    table = torch.randn(m.total_rows, 2)
    # a point exactly on a level-0 vertex: pos = x*15 + 0.5 -> x = (v - 0.5)/15 gives cell v-1.. use fraction 0
    v = torch.tensor([[4.0, 6.0, 9.0]])
    x = (v - 0.5) / 15.0
    out = tr.hashgrid_forward(x.float(), table, m)
    pos = x.float() * np.float32(15.0) + 0.5
    if torch.equal(pos, torch.floor(pos)):
        row = int(pos[0, 0]) + 16 * int(pos[0, 1]) + 256 * int(pos[0, 2])
        assert torch.allclose(out[0, :2], table[row], atol=1e-6)


def test_zero_table_zero_features_and_partition_of_unity():
    m = _meta()
    x = torch.rand(64, 3)
    assert torch.count_nonzero(tr.hashgrid_forward(x, torch.zeros(m.total_rows, 2), m)) == 0
    ones = tr.hashgrid_forward(x, torch.ones(m.total_rows, 2), m)
    assert torch.allclose(ones, torch.ones_like(ones), atol=1e-5)   # trilinear weights sum to 1


def test_rows_in_range():
    m = _meta()
    x = torch.rand(256, 3) * 1.4 - 0.2                 # includes points outside [0,1]
    rows, fr = tr.hashgrid_indices(x, m)
    for l, lvl in enumerate(m.levels):
        assert int(rows[:, l].min()) >= lvl.offset and int(rows[:, l].max()) < lvl.offset + lvl.size


def test_oneblob_sums_to_one_and_is_local():
    x = torch.rand(100, 3)
    pe = tr.oneblob_forward(x, 16).reshape(100, 3, 16)
    assert torch.allclose(pe.sum(-1), torch.ones(100, 3), atol=1e-5)
    assert (pe >= -1e-6).all()
    assert ((pe > 1e-6).sum(-1) <= 3).all()              # quartic kernel of radius 1/16 touches <= 3 bins


def test_hashgrid_input_gradient_matches_finite_difference():
    # float64 so a 1e-6 stencil practically never straddles a cell boundary (trilinear = linear along one axis)
    m = tr.grid_meta(12, 64)
    g = torch.Generator().manual_seed(5)
    table = torch.randn(m.total_rows, 2, dtype=torch.float64, generator=g)
    x = (torch.rand(8, 3, dtype=torch.float64, generator=g) * 0.8 + 0.1).requires_grad_(True)
    w = torch.randn(32, dtype=torch.float64, generator=g)
    (tr.hashgrid_forward(x, table, m) * w).sum().backward()
    eps = 1e-6
    for a in range(3):
        xp, xm = x.detach().clone(), x.detach().clone()
        xp[:, a] += eps
        xm[:, a] -= eps
        fd = ((tr.hashgrid_forward(xp, table, m) - tr.hashgrid_forward(xm, table, m)) * w).sum(-1) / (2 * eps)
        assert torch.allclose(fd, x.grad[:, a], rtol=1e-5, atol=1e-5)


def test_mlp_matches_manual_matmul():
    g = torch.Generator().manual_seed(0)
    p = tr.mlp_init(80, 33, 32, 1, g)
    x = torch.randn(10, 80, generator=g)
    W0 = p[:32 * 80].reshape(32, 80)
    W1 = p[32 * 80:].reshape(48, 32)[:33]
    assert torch.allclose(tr.mlp_forward(x, p, 80, 33, 32, 1), torch.relu(x @ W0.t()) @ W1.t(), atol=1e-6)
    assert tr.mlp_param_count(80, 33, 32, 1) == p.numel() == 32 * 80 + 48 * 32
    p2 = tr.mlp_init(112, 8, 64, 2, g)
    assert p2.numel() == 64 * 112 + 64 * 64 + 16 * 64
