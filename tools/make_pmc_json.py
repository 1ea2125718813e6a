#!/usr/bin/env python3
"""profiles/r02_pmc_traffic.json for bench.py's `roofline.traffic` from the per-configuration summaries of
tools/profile_round.sh (FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc passes, KiB).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of 16-B-per-lane streaming
reads -> doubled for the kernels whose operand loads are 16 B per lane (mlp_fwd16, gemm_lds); kernels that
read 4 B per lane (posatt_rows_*: value rows as dwords) are outside the calibration: raw value, x2 alongside.
   python tools/make_pmc_json.py <profile_round outdir> profiles/r02_pmc_traffic.json"""
import json, os, sys

src, dst = sys.argv[1], sys.argv[2]
want = {  # bench key -> (config, kernel-name prefix, loads are 16 B per lane)
    "mlp_fwd_b8": ("darcy8", "mlp_fwd16_kernel<64, 12>", True),
    "mlp_fwd_b256": ("darcy256", "gemm_lds_kernel<128, true, true, 2, false>", True),
    "posatt_rows_fwd_b8": ("darcy8", "posatt_rows_kernel<1, 0, false, false", False),
    "posatt_bwd_pair_dw_b8": ("darcy8", "posatt_bwd_pair_dw_kernel<false, false>", False),
    "posatt_rows_fwd_b256": ("darcy256", "posatt_rows_tiles<4, 1, 0, false, false>", False),
}
out = {"source": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (own passes), bench.py "
                 "--steps 20 --warmup 3; KiB per launch; see the docstring of tools/make_pmc_json.py for the x2 rule",
       "kernels": {}}
for key, (cfg, prefix, wide) in want.items():
    summ = json.load(open(os.path.join(src, cfg + ".summary.json")))
    for k in summ["kernels"]:
        if k["kernel"].startswith(prefix) and "hbm_kib_per_launch" in k:
            h = k["hbm_kib_per_launch"]
            fetch = h["fetch_x2"] if wide else h["fetch_raw"]
            out["kernels"][key] = {"kernel": k["kernel"], "launches_per_step": k["launches"], "mean_us": k["mean_us"],
                                   "fetch_kib_raw": h["fetch_raw"], "fetch_kib_x2": h["fetch_x2"], "write_kib": h["write"],
                                   "fetch_rule": "x2 (16 B per lane)" if wide else "raw (4 B per lane: uncalibrated; x2 alongside)",
                                   "traffic_bytes": int((fetch + h["write"]) * 1024),
                                   "mfma_busy_frac": k.get("mfma_busy_frac"), "valu_per_mfma": k.get("valu_per_mfma")}
            break
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
