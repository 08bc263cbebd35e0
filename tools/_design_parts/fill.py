#!/usr/bin/env python3
"""One-off (round 5): replace the @@PLACEHOLDERS@@ of DESIGN.md / README.md with the numbers of profiles/r5_*."""
import csv
import json
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(R, "profiles")
b = json.load(open(os.path.join(P, "r5_bench_default.json")))
roof, hb = b["roofline"], b["roofline"]["hbm_bound"]
c3 = b["configs"]["config3_hifimusic_bf16"]
vq = b["vq"]


def stats(name):
    rows = list(csv.DictReader(open(os.path.join(P, name))))
    steps = [int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"]][0] / 2
    out = {}
    for r in rows:
        k = r["Name"].split("(")[0].replace("void ", "").replace("mq::", "")
        out[k] = {"calls": int(r["Calls"]), "per_step": int(r["Calls"]) / steps, "ms_per_step": float(r["TotalDurationNs"]) / steps / 1e6,
                  "avg_us": float(r["AverageNs"]) / 1e3}
    return out, steps


c2s, _ = stats("r5_c2_fp32_kernel_stats.csv")
fam = lambda pre: sum(v["ms_per_step"] for k, v in c2s.items() if k.startswith(pre))
gemm_ms = fam(("conv_nt", "conv_wgrad", "conv_dgrad"))
cb_ms = fam("cb2d_")
tot_ms = sum(v["ms_per_step"] for v in c2s.values())
wfam, pfam = hb["families"]["winograd_transforms"], hb["families"]["pointwise_pool_upcat"]
ffam = hb["families"].get("fsq", {})
allk = roof["all_conv_kernels"]
rccl = json.load(open(os.path.join(P, "r5_rccl_forced_1rank_bench.json")))["collectives"]
vqk, _ = (None, None)
try:
    vrows = list(csv.DictReader(open(os.path.join(P, "r5_vq_kernel_stats.csv"))))
    rr = [r for r in vrows if "l2_rerank_kernel" in r["Name"]][0]
    vq_rr = float(rr["MinNs"]) / 1e3
except Exception:
    vq_rr = float("nan")
f = lambda v, n=1: f"{v:.{n}f}"
V = {
    "C2_MS": f(b["ms_per_step"], 1), "C2_KFPS": f(b["value"] / 1e3, 1), "C2_S": f(b["ms_per_step"] / 1e3, 4),
    "C2_DIRECT_MS": f(b["other_engines"]["fp32_direct"]["ms_per_step"], 0), "X3_MS": f(b["other_engines"]["fp32x3"]["ms_per_step"], 1),
    "C2_BF16_MS": f(b["other_engines"]["bf16"]["ms_per_step"], 1), "C2_STEP_TF": f(roof["step"]["TFLOP/s"], 1), "C2_STEP_FRAC": f(roof["step"]["frac"], 2),
    "NT4_TF": f(roof["achieved"], 1), "NT4_FRAC": f(roof["frac"], 2), "NT4_MS": f(roof["avg_launch_ms"], 3),
    "NT4_ROC_MS": f(c2s["conv_nt_kernel<128, false, 4>"]["avg_us"] / 1e3, 3), "NT4_CALLS": str(c2s["conv_nt_kernel<128, false, 4>"]["calls"]),
    "NT4_GB": f(roof["traffic"] / 1e9, 2) if roof["traffic"] else "n/a", "NT3_TF": f(allk["conv_nt_kernel<128, false, 3>"]["TFLOPs"], 0),
    "WG4_TF": f(allk["conv_wgrad_kernel<128, 128, 4>"]["TFLOPs"], 0), "PAR_TF": f(allk["conv_dgrad_parity_kernel<128, 0>"]["TFLOPs"], 0),
    "WIN_IN_US": f(c2s["winon_input_kernel<4, 5>"]["avg_us"], 0), "WIN_OUT_US": f(c2s["winon_output_kernel<4, 5>"]["avg_us"], 0),
    "WIN_DY_US": f(c2s["winon_dy_fused_kernel<4, 5>"]["avg_us"], 0), "WIN_S2D_US": f(c2s["winon_input_s2d_kernel<4, 5>"]["avg_us"], 0),
    "WIN_D2S_US": f(c2s["winon_output_d2s_kernel<4, 5>"]["avg_us"], 0),
    "H2S_PF": f(c3["roofline"]["achieved"] / 1e3, 2), "H2S_FRAC": f(c3["roofline"]["frac"], 2), "H2S_MS": f(c3["roofline"]["avg_launch_ms"], 3),
    "C3_MS": f(c3["ms_per_step"], 1), "C3_STEP_FRAC": f(c3["roofline"]["step"]["frac"], 2),
    "VQ1_US": f(vq["l2_argmin_screened_1024x256"]["ms"] * 1e3, 0), "VQ2_US": f(vq["l2_argmin_screened_2048x512"]["ms"] * 1e3, 0),
    "VQ3_US": f(vq["l2_argmin_screened_8192x512"]["ms"] * 1e3, 0),
    "VQF1_US": f(vq["l2_argmin_1024x256"]["ms"] * 1e3, 0), "VQF2_US": f(vq["l2_argmin_2048x512"]["ms"] * 1e3, 0), "VQF3_US": f(vq["l2_argmin_8192x512"]["ms"] * 1e3, 0),
    "VQ1_FRAC": f(vq["l2_argmin_screened_1024x256"]["frac"], 3), "VQ2_FRAC": f(vq["l2_argmin_screened_2048x512"]["frac"], 3),
    "VQ3_FRAC": f(vq["l2_argmin_screened_8192x512"]["frac"], 3), "VQ_RR_US": f(vq_rr, 0),
    "FSQ_TBS": f(vq["fsq"]["achieved_GBps"] / 1e3, 2), "FSQ_FRAC": f(vq["fsq"]["frac"], 2),
    "HB_W_L": f(wfam["launches_per_step"], 0), "HB_W_MS": f(wfam["ms_per_step"], 1), "HB_W_GB": f(wfam["GB_per_step"], 0),
    "HB_W_TBS": f(wfam["achieved"], 2), "HB_W_FRAC": f(wfam["frac"], 2),
    "HB_P_L": f(pfam["launches_per_step"], 0), "HB_P_MS": f(pfam["ms_per_step"], 1), "HB_P_GB": f(pfam["GB_per_step"], 0),
    "HB_P_TBS": f(pfam["achieved"], 2), "HB_P_FRAC": f(pfam["frac"], 2),
    "HB_F_L": f(ffam.get("launches_per_step", 0), 0), "HB_F_MS": f(ffam.get("ms_per_step", 0), 3), "HB_F_GB": f(ffam.get("GB_per_step", 0) or 0, 4),
    "WASTE": f(roof["wasted_traffic_ratio"], 1) + "×", "WASTE_MOVED": f(hb["winograd2d_layers"]["moved_GB_per_step"], 0),
    "WASTE_ALG": f(hb["winograd2d_layers"]["algorithmic_GB_per_step"], 1),
    "FAM_GEMM_MS": f(gemm_ms, 1), "FAM_CB_MS": f(cb_ms, 1),
    "FAM_REST_MS": f(tot_ms - gemm_ms - cb_ms - wfam["ms_per_step"] - pfam["ms_per_step"], 1),
    "LOADER_MS": f(b["configs"]["config2_through_the_loader"]["ms_per_step"], 1), "HOST_MS": f(b["host_ms_per_step"], 0),
    "REENC_ORDER": f(b["configs"]["reencode_n2_hifispeech_fp32"]["order"]["value"] / 1e3, 0),
    "REENC_SORTED": f(b["configs"]["reencode_n2_hifispeech_fp32"]["sorted"]["value"] / 1e3, 0),
    "CPU_FPS": f(b["cpu_baseline"]["value"], 1), "CPU_CORES": str(b["cpu_baseline"]["cores"]),
    "TAIL_D": f(rccl["ms_between_last_bucket_ready_and_clip"]["discriminators"], 2), "TAIL_G": f(rccl["ms_between_last_bucket_ready_and_clip"]["generator"], 2),
    "LAST_D": f(rccl["last_bucket_MiB"]["discriminators"], 2), "LAST_G": f(rccl["last_bucket_MiB"]["generator"], 2),
    "N_GPU_TESTS": sys.argv[1] if len(sys.argv) > 1 else "355",
}
for name in ("DESIGN.md", "README.md"):
    path = os.path.join(R, name)
    t = open(path).read()
    missing = sorted(set(re.findall(r"@@([A-Z0-9_]+)@@", t)) - set(V))
    if missing:
        print(name, "unknown placeholders:", missing)
    for k, v in V.items():
        t = t.replace("@@" + k + "@@", v)
    open(path, "w").write(t)
    print(name, "left:", sorted(set(re.findall(r"@@([A-Z0-9_]+)@@", t))))
