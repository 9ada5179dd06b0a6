"""Checkpoint interop with the reference's ``model.pt`` files.

What the reference writes (slams/mapping.py:1119-1128 through models/checkpoint.py:21-35) is one ``torch.save`` dict:
``decoder`` = ``Decoder.state_dict()`` -- five flat fp32 tinycudann ``params`` tensors under the keys
``pe_fn.grid_fn.params``, ``coarse_fn.decoder.params``, ``out_fn.color_decoder.params``, ``out_fn.logit_decoder.params``,
``merge.decoder.params`` -- plus ``fine_decoders`` (per-class networks), ``keyframe_dict``, ``keyframe_list``,
``estimate_c2w_list``, ``gt_c2w_list``, ``scene``, ``idx``.  This module's modules carry the same keys, so the file
layout is shared; the real interop problem is INSIDE the flat tensors, and that is what this file handles:

* MLP ``params`` (tcnn CutlassMLP / FullyFusedMLP): row-major ``[W_in (n x in) | W_hidden (n x n)... | W_out (out_pad x n)]``
  with the output rows padded to the tensor-core granule -- 8 for CutlassMLP (the reference's otype, models/decoder.py:60),
  16 for FullyFusedMLP; the kernels here use 16.  ``repack_mlp_params`` converts between paddings (padded rows carry no
  information: zero-filled, never read back), so a CutlassMLP-written tensor of ``n*in + (L-1)*n*n + pad8(out)*n`` elements
  loads into the ``pad16`` layout and a file written here can be exported with either padding.
* Hash-grid ``params``: ``[sum_l size_l, F]`` rows, levels concatenated in order; sizes follow from the level table.  tcnn
  evaluates the level scale with CUDA's float32 ``exp2f`` where this build rounds the float64 value once (DESIGN.md section
  2): both give the same table for the reference's scenes unless CUDA's last bit flips the finest level's resolution; a
  size mismatch is reported as such (no silent truncation).
* ``fine_decoders``: the reference pickles tinycudann module objects (``tinycudann.modules.Network`` instances, whose
  pickled state is their ``__dict__`` minus the native handle).  No AMD box has tinycudann, so ``load`` unpickles with
  ``_RefUnpickler``: every class under the ``tinycudann`` package resolves to a plain ``nn.Module`` stand-in that just
  receives the pickled ``__dict__`` -- its ``_parameters['params']`` is the flat tensor wanted.  Written here as
  ``{class_id: flat params}``; on load both forms are accepted (tensors, or objects exposing ``.params``).
This is the limit of what can be verified offline: tinycudann is absent from /root/reference, so the layout
facts above are the published ones, and ``tests/test_host_logic.py`` pins the repacking arithmetic, not a tcnn-written file.
"""
from __future__ import annotations

import os
import pickle
import types
from typing import Dict, Optional

import torch
from torch import nn

MLP_KEYS = ("coarse_fn.decoder.params", "out_fn.color_decoder.params", "out_fn.logit_decoder.params", "merge.decoder.params")


def _pad(n: int, g: int) -> int:
    return (n + g - 1) // g * g


def mlp_numel(n_in: int, n_out: int, n_neurons: int, n_hidden_layers: int, granule: int) -> int:
    return n_neurons * n_in + (n_hidden_layers - 1) * n_neurons * n_neurons + _pad(n_out, granule) * n_neurons


def repack_mlp_params(flat: torch.Tensor, n_in: int, n_out: int, n_neurons: int, n_hidden_layers: int,
                      dst_granule: int = 16) -> torch.Tensor:
    """Flat tcnn MLP parameters with ANY output-row padding (8 or 16, inferred from the element count) -> the same
    network with ``dst_granule`` padding.  Raises ValueError when the count matches no padding of this shape."""
    body = n_neurons * n_in + (n_hidden_layers - 1) * n_neurons * n_neurons
    rows_src = (flat.numel() - body) // n_neurons if flat.numel() > body else -1
    if rows_src < n_out or body + rows_src * n_neurons != flat.numel() or rows_src not in (_pad(n_out, 8), _pad(n_out, 16), n_out):
        raise ValueError(f"MLP params of {flat.numel()} elements do not describe a {n_in}->{n_neurons}x{n_hidden_layers}->{n_out} "
                         f"network (expected {mlp_numel(n_in, n_out, n_neurons, n_hidden_layers, 8)} with 8-row or "
                         f"{mlp_numel(n_in, n_out, n_neurons, n_hidden_layers, 16)} with 16-row output padding)")
    out = flat.new_zeros(body + _pad(n_out, dst_granule) * n_neurons)
    out[:body] = flat[:body]
    out[body:body + n_out * n_neurons] = flat[body:body + n_out * n_neurons]
    return out


class _TcnnModuleStandIn(nn.Module):
    """What a pickled ``tinycudann.modules.{Module,Network,Encoding,NetworkWithInputEncoding}`` becomes on a box without
    tinycudann: an ``nn.Module`` shell holding the pickled attributes (``_parameters['params']``, ``n_input_dims``,
    ``network_config``, ...).  It computes nothing."""

    def __setstate__(self, state):
        self.__dict__.update(state)

    def forward(self, *a, **k):
        raise RuntimeError("tinycudann object from a reference checkpoint: parameters only, not callable on this build")


class _RefUnpickler(pickle.Unpickler):
    """find_class: anything under the ``tinycudann`` package -> ``_TcnnModuleStandIn`` (unless a real tinycudann with that
    class is importable); everything else as usual."""

    def find_class(self, module, name):
        if module.split(".")[0] == "tinycudann":
            try:
                return super().find_class(module, name)
            except Exception:
                return _TcnnModuleStandIn
        return super().find_class(module, name)


# the pickle_module torch.load takes: .Unpickler (torch subclasses it) and .load (legacy, non-zip files)
_ref_pickle = types.ModuleType("dns_slam_amd._ref_pickle")
_ref_pickle.Unpickler = _RefUnpickler
_ref_pickle.load = lambda f, **kw: _RefUnpickler(f, **kw).load()
_ref_pickle.__dict__.update({k: getattr(pickle, k) for k in ("dumps", "dump", "loads", "Pickler", "PickleError", "UnpicklingError",
                                                             "HIGHEST_PROTOCOL", "DEFAULT_PROTOCOL")})


def _net_shape(net):
    return net.n_input_dims, net.n_output_dims, net.n_neurons, net.n_hidden_layers


class Checkpoint:
    """``Checkpoint(dir, device=..., decoder=dec, fine_decoders=mapper.fine_decoders)``; ``save(name, **state)`` /
    ``load(name) -> remaining state`` as the reference's class is used (slams/mapping.py:1119-1128, dns_slam.py:155-159)."""

    def __init__(self, checkpoint_dir: str = "./chkpts", device=None, decoder=None, fine_decoders=None, **modules):
        self.dir, self.device = checkpoint_dir, device
        self.decoder, self.fine_decoders, self.modules = decoder, fine_decoders, modules
        os.makedirs(checkpoint_dir, exist_ok=True)

    def _path(self, name: str) -> str:
        return name if os.path.isabs(name) else os.path.join(self.dir, name)

    # ------------------------------------------------------------------ write
    def save(self, name: str, mlp_granule: int = 16, **state) -> str:
        """mlp_granule = 8 writes the MLP tensors with CutlassMLP's padding (what a tinycudann build of the reference
        expects), 16 the native layout."""
        blob = dict(state)
        if self.decoder is not None:
            sd = {k: v.detach().cpu().clone() for k, v in self.decoder.state_dict().items()}
            if mlp_granule != 16:
                for key in MLP_KEYS:
                    if key in sd:
                        sd[key] = repack_mlp_params(sd[key], *_net_shape(self._module_of(key)), dst_granule=mlp_granule)
            blob["decoder"] = sd
        if self.fine_decoders is not None:
            pool = self.fine_decoders
            shape = (pool.n_in, pool.n_out, pool.nn_, pool.nl)
            blob["fine_decoders"] = {int(c): repack_mlp_params(pool.params_of(c).detach().cpu(), *shape, dst_granule=mlp_granule)
                                     for c in pool.keys()}
        for key, mod in self.modules.items():
            blob[key] = mod.state_dict()
        path = self._path(name)
        torch.save(blob, path)
        return path

    def _module_of(self, key: str):
        mod = self.decoder
        for part in key.split(".")[:-1]:
            mod = getattr(mod, part)
        return mod

    # ------------------------------------------------------------------ read
    def load(self, name: str) -> Dict:
        blob = torch.load(self._path(name), map_location=self.device or "cpu", weights_only=False, pickle_module=_ref_pickle)
        if self.decoder is not None and "decoder" in blob:
            own = self.decoder.state_dict()
            for key, src in blob.pop("decoder").items():
                if key not in own:
                    continue
                src = src.detach().to(own[key].dtype).reshape(-1)
                if key in MLP_KEYS and src.numel() != own[key].numel():
                    src = repack_mlp_params(src, *_net_shape(self._module_of(key)))
                if src.numel() != own[key].numel():
                    raise ValueError(f"checkpoint tensor '{key}' has {src.numel()} elements, this model {own[key].numel()}"
                                     + (" (hash-grid level table differs: other bound / voxel_size / hash_size?)"
                                        if key.endswith("grid_fn.params") else ""))
                own[key] = src.reshape(own[key].shape)
            self.decoder.load_state_dict(own)
        if self.fine_decoders is not None and "fine_decoders" in blob:
            pool = self.fine_decoders
            shape = (pool.n_in, pool.n_out, pool.nn_, pool.nl)
            for c, net in dict(blob.pop("fine_decoders")).items():
                flat = net if torch.is_tensor(net) else getattr(net, "params", None)
                if flat is None:
                    raise ValueError(f"fine decoder of class {c}: neither a tensor nor an object with .params")
                pool.add(int(c))
                with torch.no_grad():
                    pool.params_of(int(c)).copy_(repack_mlp_params(flat.detach().float().reshape(-1).cpu(), *shape).to(pool.pool.device))
        for key, mod in self.modules.items():
            if key in blob:
                mod.load_state_dict(blob.pop(key))
        return blob


def load_reference_decoder(path: str, decoder, fine_decoders=None, device: Optional[str] = None) -> Dict:
    """One-call form: read a reference-written ``model.pt`` into ``decoder`` (+ the per-class pool); returns the rest."""
    d, n = os.path.split(os.path.abspath(path))
    return Checkpoint(d, device=device, decoder=decoder, fine_decoders=fine_decoders).load(n)
