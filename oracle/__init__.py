"""CPU oracle for the DNS-SLAM volumetric-rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``dns_slam_amd/`` (the product) may
import, call, link or execute anything from this package.  The only legal
importers are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- and there only as the checker / the timed CPU baseline,
never as the thing shipped.

What it is: a plain PyTorch-CPU (fp32/fp64) restatement of the reference's
algorithm for every row of SURVEY.md section 8(a):

* ``render_math``  -- pixel pick, ray generation, box clip, depth-guided
  sampling, occupancy compositing, free-space / opacity losses, quaternion->R
  (reference ``utils/common.py``; PINNED: checked bit-for-bit / to 1e-6 against
  outputs of the imported reference functions, committed under
  ``tests/golden/`` together with the generating script
  ``tests/golden/make_golden.py``).
* ``tcnn_ref``     -- OneBlob, multi-resolution hash grid, bias-free ReLU MLP.
  These live in the third-party, un-vendored CUDA package ``tinycudann``
  (reference ``requirements.txt:35``, unpinned git HEAD; fallback commit
  91ee479d275d322a65726435040fc20b56b9c991 named in reference ``README.md:57``).
  The reference holds no test, golden vector or fixture for them and the package
  cannot be built here (CUDA only): PARITY UNPINNED for these three functions --
  they restate tiny-cuda-nn's published algorithm (Mueller et al. 2022) and are
  anchored only by the reference's own call sites
  (``models/pos_encoding.py:31-71``, ``models/decoder.py:58-116``,
  ``slams/mapping.py:737``) and by known-answer tests.
* ``slam_ref``     -- ``Mapper.renderer`` / ``Tracker.renderer`` /
  ``get_target_samples`` / the seven mapping losses / the three tracking losses /
  one optimise iteration (reference ``slams/mapping.py``, ``slams/tracking.py``;
  these modules cannot be imported here -- missing cv2/colorama/tinycudann --
  so they are pinned through the ``utils/common.py`` functions they call plus the
  head wiring restated from the cited lines).
"""
