"""CPU: the product's host-side level table (C, libm float32) equals the oracle's (numpy float32) bit for bit on
every scene bound the reference ships and on the benchmark configs; known sizes from SURVEY Appendix A7."""
import numpy as np
import pytest
import torch

from dns_slam_amd import ops, synthetic
from oracle import tcnn_ref as tr

REPLICA = {  # reference configs/replica/*.yaml back_end.bound
    "room_0": [[-2.9, 8.9], [-3.2, 5.5], [-3.5, 3.3]],
    "room_1": [[-7.0, 2.8], [-4.6, 4.3], [-3.0, 2.9]],
    "office_0": [[-5.5, 5.9], [-6.7, 5.4], [-4.7, 5.3]],
    "scene0000": [[-0.1, 8.6], [-0.1, 8.9], [-0.3, 3.3]],
}


@pytest.mark.parametrize("scene,hash_size,voxel", [("room_0", 16, 0.02), ("room_1", 16, 0.02), ("office_0", 16, 0.02),
                                                   ("scene0000", 20, 0.04), ("room_0", 20, 0.04), ("room_0", 19, 0.01)])
def test_level_table_matches_oracle(scene, hash_size, voxel):
    bound = synthetic.load_bound(REPLICA[scene])
    res = tr.desired_resolution(bound, voxel)
    om = tr.grid_meta(hash_size, res)
    pm = ops.GridMeta(hash_size, res)
    assert pm.total_rows == om.total_rows
    for lo, lp in zip(om.levels, pm.levels()):
        assert np.float32(lo.scale).tobytes() == np.float32(lp["scale"]).tobytes()
        assert (lo.resolution, lo.size, lo.offset, lo.hashed) == (lp["resolution"], lp["size"], lp["offset"], lp["hashed"])


def test_room0_table_size():
    bound = synthetic.load_bound(REPLICA["room_0"])
    m = tr.grid_meta(16, tr.desired_resolution(bound, 0.02))
    assert m.total_rows == 853312                       # 6.83 MB of fp32 pairs
    assert [l.resolution for l in m.levels[:4]] == [16, 21, 26, 33]
    assert [l.hashed for l in m.levels] == [False] * 4 + [True] * 12
    assert all(l.size == 65536 for l in m.levels[4:])


def test_grid_meta_pickles():
    import pickle
    pm = ops.GridMeta(16, 592)
    q = pickle.loads(pickle.dumps(pm))
    assert q.total_rows == pm.total_rows and q.levels() == pm.levels()
