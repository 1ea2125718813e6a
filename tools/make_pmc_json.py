#!/usr/bin/env python3
"""profiles/r06_pmc_traffic.json for bench.py's `roofline.traffic` from the per-configuration summaries of
tools/profile_round.sh (FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc passes, KiB).
Correction (profiles/r03_fetch_calibration.txt, tools/fetch_calib.sh): on gfx950 FETCH_SIZE reports exactly HALF of the
bytes for every coalesced read shape these kernels use (2 / 4 / 16 B per lane, row gathers, buffer loads alike - not only
the 16-B streaming reads MI355X_MICROARCH.md calibrates), WRITE_SIZE is exact: traffic = 2 x FETCH_SIZE + WRITE_SIZE.
   python tools/make_pmc_json.py <profile_round outdir> profiles/r06_pmc_traffic.json"""
import json, os, sys

src, dst = sys.argv[1], sys.argv[2]
want = {  # bench key -> (config, kernel-name prefix, shape of the launch(es) the counters were collected on - bench.py only
         # quotes the bytes for a probe of exactly this shape; "family": the kernel runs on several shapes in the step and the
         # figure is their mean, which no single-shape probe may quote)
    "block_bwd_kernel_b8": ("darcy8", "block_bwd_kernel<2, 2>", "L256_H2_D64_b8"),
    "block_fwd_kernel_b8": ("darcy8", "block_fwd_kernel<2, true>", "L256_H2_D64_b8"),      # (<H, WLDS>: weights and E rows through LDS)
    # round 5: the fused encoder- / decoder-side launches of the small regime (csrc/pit_edge.hip)
    "decoder_fwd_kernel_b8": ("darcy8", "decoder_fwd_kernel<2, 64, true>", "N1849_J256_H2_D64_b8"),
    "decoder_bwd_kernel_b8": ("darcy8", "decoder_bwd_kernel<2, 64, true>", "N1849_J256_H2_D64_b8"),
    "encoder_fwd_kernel_b8": ("darcy8", "encoder_fwd_kernel<2, 64, 4>", "N256_J1849_H2_D64_b8"),
    "encoder_bwd_kernel_b8": ("darcy8", "encoder_bwd_kernel<2, 64, 4>", "N256_J1849_H2_D64_b8"),
    "decoder_fwd_kernel_b256": ("darcy256", "decoder_fwd_kernel<2, 64, true>", "N1849_J256_H2_D64_b256"),
    "decoder_bwd_kernel_b256": ("darcy256", "decoder_bwd_kernel<2, 64, true>", "N1849_J256_H2_D64_b256"),
    "mlp_fwd_b256": ("darcy256", "mlp_fwd64_kernel<12, false>", "rows65536_192_64_64"),          # the four processor MLPs (one shape)
    "posatt_rows_fwd_b256": ("darcy256", "posatt_rows_tiles<4, 1, 0, false, false, true>", "256x256_D64_H2_b256"),
    "mlp_dw_b256": ("darcy256", "gemm_rr_kernel<1, 1, 64, false>", "family"),
}
out = {"source": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (own passes), bench.py "
                 "--steps 20 --warmup 3; KiB per launch; traffic_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                 "(profiles/r03_fetch_calibration.txt)",
       "kernels": {}}
for key, (cfg, prefix, shape) in want.items():
    if not os.path.exists(os.path.join(src, cfg + ".summary.json")):
        continue
    summ = json.load(open(os.path.join(src, cfg + ".summary.json")))
    for k in summ["kernels"]:
        if k["kernel"].startswith(prefix) and "hbm_kib_per_launch" in k:
            h = k["hbm_kib_per_launch"]
            out["kernels"][key] = {"kernel": k["kernel"], "shape": shape, "launches_per_step": k["launches"], "mean_us": k["mean_us"],
                                   "fetch_kib_raw": h["fetch_raw"], "fetch_kib_x2": h["fetch_x2"], "write_kib": h["write"],
                                   "traffic_bytes": int((h["fetch_x2"] + h["write"]) * 1024),
                                   "mfma_busy_frac": k.get("mfma_busy_frac"), "valu_per_mfma": k.get("valu_per_mfma")}
            break
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
