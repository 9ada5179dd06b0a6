"""``MapStep`` (dns_slam_amd/fused_step.py: the mapping iteration as a fixed launch sequence over preallocated buffers) against
the autograd-driven iteration of ``Mapper.optimize_frames`` (itself held to the oracle by test_gpu_slam.py / test_gpu_cfg1.py):
same seed -> same draws -> the same losses every iteration and the same parameters after several Adam steps."""
import copy

import pytest
import torch

from test_gpu_slam import _setup
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(fused, code, n_iters, layout, nn=32, nl=1):
    from dns_slam_amd.fused_step import MapStep
    from dns_slam_amd.optim import FusedAdam  # noqa: F401  (set_optimizer(fused=True))
    cfg, bound, cam, frames, dec, mapper = _setup(nn, nl, layout=layout)
    mapper.static_shapes, mapper.is_BA, mapper.overlap_smooth, mapper.prefetch_draws = True, True, True, True
    opt, ql, Tl = mapper.set_optimizer(frames, fused=True)
    for grp, lr in zip(opt.param_groups, (mapper.lr, mapper.BA_cam_lr, mapper.BA_cam_lr)):
        grp["lr"] = lr
    prep = mapper.prepare_frames(frames)
    feats = None
    if code:
        g = torch.Generator().manual_seed(5)
        npf = prep["n1"] + prep["n2"]
        feats = (torch.rand(4 * npf, 32 + 15, 32, generator=g) * 2 - 1).to(DEV)
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    hist, grads = [], None
    pool = mapper.fine_decoders.pool
    if fused:
        ms = MapStep(mapper, frames, ql, Tl, prep=prep, features=feats)
        for i in range(n_iters):
            ms.step()
            total, terms = ms.losses()
            hist.append((float(total), {k: float(v) for k, v in terms.items()}))
            if i == 0:
                grads = {"table": ms.g_table, "coarse": ms.g_coarse, "color": ms.g_color, "logit": ms.g_logit,
                         "pool": ms.g_pool.view_as(pool), "quat": ms.g_quat.view(4, 4)[1:], "trans": ms.g_trans.view(4, 3)[1:]}
                grads = {k: v.detach().cpu().clone() for k, v in grads.items()}
        ms.write_back()
    else:
        for i in range(n_iters):
            opt.zero_grad(set_to_none=True)
            s = mapper.get_target_samples(frames, ql, Tl, prep=prep, features=feats)
            loss, terms = mapper.iteration_loss(s, lambda_lt=10.0, smooth=True)
            loss.backward()
            if i == 0:
                grads = {"table": dec.pe_fn.grid_fn.params.grad, "coarse": dec.coarse_fn.decoder.params.grad,
                         "color": dec.out_fn.color_decoder.params.grad, "logit": dec.out_fn.logit_decoder.params.grad,
                         "pool": pool.grad, "quat": torch.stack([q.grad for q in ql[1:]]),
                         "trans": torch.stack([t.grad for t in Tl[1:]])}
                grads = {k: v.detach().cpu().clone() for k, v in grads.items()}
            opt.step()
            hist.append((float(loss.detach()), {k: float(v) for k, v in terms.items()}))
    torch.cuda.synchronize()
    params = {"table": dec.pe_fn.grid_fn.params, "coarse": dec.coarse_fn.decoder.params,
              "color": dec.out_fn.color_decoder.params, "logit": dec.out_fn.logit_decoder.params,
              "pool": mapper.fine_decoders.pool, "quat": torch.stack([q.detach() for q in ql]),
              "trans": torch.stack([t.detach() for t in Tl])}
    return hist, {k: v.detach().cpu().clone() for k, v in params.items()}, grads, (mapper.lr, mapper.BA_cam_lr)


@pytest.mark.parametrize("code,layout,net", [(False, "reference_tiled", (32, 1)), (True, "reference_tiled", (32, 1)),
                                             (True, "per_ray", (64, 2))])
def test_map_step_equals_the_autograd_iteration(code, layout, net):
    n = 6
    ha, pa, ga, lrs = _run(False, code, n, layout, *net)
    hf, pf, gf, _ = _run(True, code, n, layout, *net)
    for i, ((la, ta), (lf, tf)) in enumerate(zip(ha, hf)):
        assert abs(la - lf) <= 1e-4 * abs(la), (i, la, lf)
        for k in ta:
            assert abs(ta[k] - tf[k]) <= 1e-4 * max(abs(ta[k]), 1e-6), (i, k, ta[k], tf[k])
    for k in ga:                                                 # the first iteration's gradients, before Adam touches anything
        assert_close(gf[k], ga[k], rtol=1e-4, elementwise=False, what=f"MapStep vs autograd: d {k}, iteration 1")
    for k in pa:
        # Adam's first steps move every weight by ~lr whatever the size of its gradient: where the gradient is rounding noise
        # around zero (table rows no sample touched strongly) the two runs' steps differ by a fraction of lr.  So: every
        # parameter within 1e-4 of the tensor's scale + 5 % of the distance n Adam steps can move it
        lr = lrs[1] if k in ("quat", "trans") else lrs[0]
        diff = (pf[k] - pa[k]).abs().max().item()
        assert diff <= 1e-4 * pa[k].abs().max().item() + 0.05 * lr * n, (k, diff)
    assert ha[-1][0] < ha[0][0]                                  # and it trains


def test_map_step_frozen_poses_and_single_frame():
    """is_BA False: no pose gradient, no d(grid)/dx buffer; the poses stay bit-equal."""
    from dns_slam_amd.fused_step import MapStep
    cfg, bound, cam, frames, dec, mapper = _setup()
    mapper.static_shapes, mapper.is_BA = True, False
    _, ql, Tl = mapper.set_optimizer(frames, fused=True)
    q0 = torch.stack([q.detach().clone() for q in ql])
    torch.manual_seed(9)
    torch.cuda.manual_seed(9)
    ms = MapStep(mapper, frames, ql, Tl)
    l0 = None
    for i in range(8):
        ms.step()
        if i == 0:
            l0 = float(ms.losses()[0])
    ms.write_back()
    assert ms.dydx is None and torch.equal(torch.stack([q.detach() for q in ql]), q0)
    assert float(ms.losses()[0]) < l0
