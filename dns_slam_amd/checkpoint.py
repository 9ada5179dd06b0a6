"""Mirror of reference models/checkpoint.py:5-66 (``Checkpoint``): save the registered modules' ``state_dict()`` plus
arbitrary extra state with ``torch.save``; load merges matching keys.  The reference class is model-agnostic; what
matters for interop is the key layout it ends up writing (slams/mapping.py:1119-1128): ``decoder`` (state dict with
``pe_fn.grid_fn.params``, ``coarse_fn.decoder.params``, ``out_fn.color_decoder.params``, ``out_fn.logit_decoder.params``,
``merge.decoder.params`` -- the same flat fp32 tensors here), ``fine_decoders`` (pickled; here the pooled module, which
exposes ``keys()`` / ``[class_id]`` like the reference's dict), ``keyframe_dict``, ``keyframe_list``, pose lists."""
import os

import torch


class Checkpoint:
    def __init__(self, checkpoint_dir="./chkpts", device=None, **kwargs):
        self.module_dict = kwargs
        self.device = device
        self.checkpoint_dir = checkpoint_dir
        os.makedirs(checkpoint_dir, exist_ok=True)

    def _path(self, filename):
        return filename if os.path.isabs(filename) else os.path.join(self.checkpoint_dir, filename)

    def save(self, filename, **kwargs):
        outdict = dict(kwargs)
        for k, v in self.module_dict.items():
            outdict[k] = v.state_dict()
        torch.save(outdict, self._path(filename))

    def load(self, filename):
        state_dict = torch.load(self._path(filename), map_location=self.device, weights_only=False)
        for k, v in self.module_dict.items():
            if k in state_dict:
                model_dict = v.state_dict()
                for kk, vv in state_dict[k].items():
                    if kk in model_dict:
                        model_dict[kk] = vv
                v.load_state_dict(model_dict)
        return {k: v for k, v in state_dict.items() if k not in self.module_dict}
