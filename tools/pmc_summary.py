"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/pmc_traffic.json + a text table.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 3 --warmup 2 --no-overlap --no-cpu-baseline --no-kernel-timing --no-render-forward
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py ... (same)
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write cfg2 r02 5      # 5 = warmup + steps of the run

Units / corrections (MI355X_MICROARCH.md, HBM): counters are KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
coalesced (16 B/lane) streaming read -> doubled for the streaming kernels (MLP, losses, compositing, transpose, Adam);
8-byte gather kernels (encode_*, hashgrid_bwd_binned) are uncalibrated and left RAW (a lower bound).  WRITE_SIZE is exact."""
import collections, csv, glob, json, os, sys

fetch_dir, write_dir, workload, tag = sys.argv[1:5]
n_steps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
ENTRY = {  # C-ABI entry point -> kernels it launches
    "dns_mlp_fwd": ["mlp_fwd_kernel"], "dns_mlp_bwd": ["mlp_bwd_kernel", "mlp_dwin_kernel"],
    "dns_encode_fwd": ["encode_fwd_kernel"], "dns_encode_bwd": ["encode_bwd_kernel", "dgrid_transpose_kernel", "hashgrid_bwd_binned_kernel", "hashgrid_bwd_pairlist_kernel",
                                                                "hashgrid_bwd_pairbins_kernel", "hashgrid_bwd_partition_kernel", "hashgrid_bwd_queue_kernel"],
    "dns_composite_fwd": ["composite_fwd_kernel"], "dns_composite_bwd": ["composite_bwd_kernel"],
    "dns_loss_sums": ["loss_ray_sums_kernel", "loss_point_sums_kernel"], "dns_loss_bwd": ["loss_ray_bwd_kernel", "loss_point_bwd_kernel"],
    "dns_raygen_sample": ["depth_max_kernel", "raygen_sample_kernel"], "dns_raygen_bwd": ["raygen_bwd_reduce_kernel", "raygen_bwd_pose_kernel"],
    "dns_adam_step": ["adam_tick_kernel", "adam_step_kernel"], "dns_tv_fwd": ["tv_fwd_kernel"], "dns_tv_bwd": ["tv_bwd_kernel"],
    "dns_group_slots": ["group_hist_kernel", "group_scan_kernel", "group_scatter_kernel"],
    "dns_class_slots": ["class_slots_kernel"], "dns_feature_block": ["feature_block_kernel"], "dns_rgb_sigmoid": ["rgb_sigmoid_kernel"],
    "dns_raw_bwd": ["raw_bwd_kernel"], "dns_lattice_points": ["lattice_points_kernel"],
    # ABI v9 (round 4)
    "dns_encode_fwd_split": ["encode_fwd_split_kernel"], "dns_feature_block_split": ["feature_block_split_kernel"],
    "dns_mlp_dwin": ["mlp_dwin_kernel"], "dns_feature_gather_frames": ["feature_gather_frames_kernel"],
    "dns_merge_dy": ["merge_dy_kernel"], "dns_add_ref_sum": ["add_ref_sum_kernel"], "dns_refer_poses": ["refer_poses_kernel"],
    "dns_draw_finish": ["draw_finish_kernel"], "dns_loss_finalize": ["loss_finalize_kernel"],
    "dns_loss_rays": ["loss_point_sums_kernel", "loss_rays_fused_kernel"],
    # ABI v12 (round 5): half rows.  (x rows are 16 B / lane reads -> FETCH doubled like the other MLP kernels; their dY rows are
    # dword reads, so the doubled figure is an UPPER bound for these two)
    "dns_mlp_fwd_half": ["mlp_half_fwd_kernel"], "dns_mlp_bwd_half": ["mlp_half_bwd_kernel"],
}
GATHER = ("encode_fwd_kernel", "encode_fwd_split_kernel", "encode_bwd_kernel", "hashgrid_bwd_binned_kernel", "feature_gather_frames_kernel",
          "hashgrid_bwd_pairbins_kernel", "hashgrid_bwd_pairlist_kernel")


def load(d):
    f = (glob.glob(os.path.join(d, "*counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*counter_collection.csv")))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "dns::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0].split("::")[-1]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"]) * 1024.0
    return agg


fe, wr = load(fetch_dir), load(write_dir)
# iterations actually profiled = launches of the optimiser kernel (bench.py runs warm-up + timed + a per-step spread pass)
if "adam_step_kernel" in fe or "adam_step_kernel" in wr:
    n_steps = max(fe.get("adam_step_kernel", [0])[0], wr.get("adam_step_kernel", [0])[0]) or n_steps
kern = {}
for k in sorted(set(fe) | set(wr)):
    n = max(fe.get(k, [0])[0], wr.get(k, [0])[0]) or 1
    raw_f = fe.get(k, [0, 0.0])[1] / n
    f = raw_f if k in GATHER else 2.0 * raw_f
    w = wr.get(k, [0, 0.0])[1] / n
    kern[k] = {"launches": n, "launches_per_step": n / n_steps, "fetch_raw_bytes": raw_f, "fetch_bytes": f, "write_bytes": w,
               "fetch_corrected": k not in GATHER}
entries = {}
for e, ks in ENTRY.items():
    present = [k for k in ks if k in kern]
    if not present:
        continue
    calls = max(kern[k]["launches"] for k in present)
    tot = sum((kern[k]["fetch_bytes"] + kern[k]["write_bytes"]) * kern[k]["launches"] for k in present) / calls
    entries[e] = {"hbm_bytes_per_launch": tot, "kernels": present}
out = {"workload": workload, "source": f"profiles/{tag}_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                      "FETCH doubled for streaming kernels, raw for 8-byte gather kernels)",
       "steps_profiled": n_steps,
       "hbm_bytes_per_step": sum((v["fetch_bytes"] + v["write_bytes"]) * v["launches"] for v in kern.values()) / n_steps,
       "kernels": kern, "entry_points": entries}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
with open(f"profiles/{tag}_pmc_traffic.txt", "w") as fh:
    fh.write(__doc__ + "\n")
    fh.write(f"{'kernel':34s} {'launches':>8s} {'FETCH raw MB':>13s} {'FETCH used MB':>14s} {'WRITE MB':>10s}\n")
    for k, v in sorted(kern.items(), key=lambda kv: -(kv[1]['fetch_bytes'] + kv[1]['write_bytes'])):
        fh.write(f"{k:34s} {v['launches']:8d} {v['fetch_raw_bytes'] / 1e6:13.2f} {v['fetch_bytes'] / 1e6:14.2f} {v['write_bytes'] / 1e6:10.2f}\n")
    fh.write(f"\nall kernels of the library, per iteration: {out['hbm_bytes_per_step'] / 1e6:.1f} MB\n")
    fh.write("\nper C-ABI entry point (bytes per call):\n")
    for e, v in sorted(entries.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        fh.write(f"{e:22s} {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB   ({', '.join(v['kernels'])})\n")
print(open(f"profiles/{tag}_pmc_traffic.txt").read())
