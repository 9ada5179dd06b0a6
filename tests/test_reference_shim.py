"""INTEGRATION.md route 1 (zero-edit): the reference's own ``models/decoder.py`` / ``models/pos_encoding.py`` constructed
against ``dns_slam_amd.tcnn_shim`` registered as ``tinycudann``.  Needs the reference sources, so it runs in the build
container only (skipped on the GPU box, where /root/reference does not exist); nothing here touches a GPU: construction,
constructor arguments and the state-dict layout are what the route depends on."""
import importlib
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference sources not present")


@pytest.fixture()
def reference_models():
    from dns_slam_amd import tcnn_shim
    saved = {k: sys.modules.get(k) for k in ("tinycudann", "models", "models.decoder", "models.pos_encoding")}
    sys.modules["tinycudann"] = tcnn_shim
    for k in ("models", "models.decoder", "models.pos_encoding"):
        sys.modules.pop(k, None)
    sys.path.insert(0, REF)
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True                      # nothing is written next to the read-only sources
    try:
        yield importlib.import_module("models.decoder")
    finally:
        sys.dont_write_bytecode = old
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_reference_decoder_constructs_on_the_shim(reference_models):
    from dns_slam_amd import synthetic
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd import tcnn_shim
    bound = synthetic.load_bound(synthetic.ROOM0_BOUND)
    cfg = synthetic.default_cfg()["model"]
    ref = reference_models.Decoder(cfg, bound, n_class=40)
    own = Decoder(cfg, bound, n_class=40)
    sd_ref, sd_own = ref.state_dict(), own.state_dict()
    assert set(sd_ref) == set(sd_own) == {"pe_fn.grid_fn.params", "pe_fn.pe_fn.params", "coarse_fn.decoder.params",
                                         "out_fn.color_decoder.params", "out_fn.logit_decoder.params",
                                         "merge.decoder.params", "merge.pe_fn.params"}
    for k in sd_ref:
        assert sd_ref[k].shape == sd_own[k].shape and sd_ref[k].dtype == torch.float32, k
    # every native object the reference built is the shim's, with the constructor arguments of the cited call sites
    assert isinstance(ref.coarse_fn.decoder, tcnn_shim.Network) and isinstance(ref.pe_fn.grid_fn, tcnn_shim.Encoding)
    assert (ref.coarse_fn.decoder.n_input_dims, ref.coarse_fn.decoder.n_output_dims) == (80, 33)      # decoder.py:84
    assert (ref.out_fn.color_decoder.n_input_dims, ref.out_fn.color_decoder.n_output_dims) == (112, 3)  # decoder.py:101
    assert ref.out_fn.logit_decoder.n_output_dims == 40                                                 # decoder.py:110
    assert ref.merge.decoder.n_input_dims == 112 and ref.merge.decoder.n_output_dims == 32              # decoder.py:58
    assert ref.pe_dim == 48 and ref.grid_dim == 32 and ref.pe_fn.resolution == own.pe_fn.resolution == 592
    assert ref.pe_fn.grid_fn.meta.total_rows == 853312                                                  # room_0 table
    # a state dict written by one loads into the other (checkpoint interop at the module level)
    own.load_state_dict(sd_ref)
    ref.load_state_dict(sd_own)
