#!/usr/bin/env python3
"""Fill the measured numbers into DESIGN.md / README.md:  python3 tools/render_docs.py [tag]

tools/templates/*.in hold the documents with @@NAME@@ tokens; the values come from the round-end measurement set under
profiles/ (<tag>_final_bench*.json, <tag>_final_kernel_stats.csv, <tag>_final_per_layer.txt, <tag>_sq_counters.json), so the
documents can be regenerated after every measurement pass.  Edit the templates, not the rendered files."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = os.path.join(ROOT, "profiles")


def line(name):
    txt = open(os.path.join(P, f"{tag}_final_{name}.json")).read().strip().splitlines()
    return json.loads(txt[-1])


b = line("bench")
v = {}
v["VALUE"] = f"{b['value']:.1f}"
v["MS"] = f"{b['ms_per_step']:.2f}"
fl = b.get("flops_per_step") or b.get("config", {}).get("flops_per_step")
whole = b.get("whole_step_tflops")
if whole is None and fl:
    whole = fl / (b["ms_per_step"] * 1e-3) / 1e12
if whole is None:
    whole = 9231.0 / b["ms_per_step"]            # nominal GFLOP of one bs-8 step / ms = TFLOP/s
v["WHOLE"] = f"{whole:.0f}"
v["WHOLEFRAC"] = f"{whole / 2500 * 100:.0f}"
r = b["roofline"]
v["DOMTF"] = f"{r['achieved'] / 1000:.2f}"
v["DOMFRAC"] = f"{r['frac'] * 100:.1f}"
v["DOMUS"] = f"{r.get('avg_launch_ms', 0) * 1000:.1f}" if r.get("avg_launch_ms") else "?"
v["TRAFFIC"] = f"{r['traffic'] / 1e6:.0f}" if r.get("traffic") else "n/a"
h = b.get("hbm_leg") or {}
v["HBMMS"] = f"{h.get('ms_per_step', 0):.2f}"
v["HBMGBS"] = f"{h.get('achieved', h.get('gbs', 0)) / 1000:.2f}" if (h.get("achieved") or h.get("gbs")) else "?"
c = b.get("cpu_baseline") or {}
v["CPU"] = f"{c.get('value', 0):.2f}"
v["INF32"] = f"{line('bench_infer_f32')['value']:.1f}"
v["INBF"] = f"{line('bench_infer_bf16')['value']:.0f}"
v["T1024"] = f"{line('bench_1024')['value']:.1f}"
v["UNET"] = f"{line('bench_unet')['value']:.1f}"
v["RGB"] = f"{line('bench_rgb')['value']:.1f}"
v["TF32"] = f"{line('bench_train_f32')['value']:.1f}"
v["RCCL"] = f"{line('bench_rccl1')['ms_per_step']:.2f} vs {line('bench_nocpu')['ms_per_step']:.2f}"
q = line("bench_quantify")
v["QUANT"] = f"{q['value']:.0f}"
qc = q.get("cpu_baseline") or {}
v["QCPU"] = f"{qc.get('value', 0):.2f}"

# kernel trace
rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_final_kernel_stats.csv"))))
NS = 11
v["LAUNCH"] = f"{sum(int(x['Calls']) for x in rows) / NS:.0f}"
dom = [x for x in rows if "igemm_lattice_wide_kernel<1>" in x["Name"]]
v["DOMUSPROF"] = f"{float(dom[0]['AverageNs']) / 1e3:.1f}" if dom else "?"

# per entry point
per = []
seen = False
for ln in open(os.path.join(P, f"{tag}_final_per_layer.txt")):
    if ln.startswith("---- per entry point"):
        seen = True
        continue
    if seen and ln.startswith("unetdc_"):
        f = ln.split()
        per.append(f"`{f[0][7:]}` {float(f[1]):.2f}")
v["PERLAYER"] = ", ".join(per) + "."

# SQ counters of this round
sq = json.load(open(os.path.join(P, f"{tag}_sq_counters.json")))
out = [f"Round-3 counters (`{tag}_sq_counters.json`, same passes):", "",
       "| kernel (op_bench arguments) | MFMA busy | LDS instr / MFMA | VALU instr / MFMA | waiting (`SQ_WAIT_ANY` / wave cycles) |", "|---|---|---|---|---|"]
for name, rec in sq.items():
    for k, kv in rec["kernels"].items():
        cn = kv["counters"]
        if cn.get("SQ_INSTS_MFMA", 0) <= 0:
            continue
        kk = re.sub(r"\(unetdc::.*", "", k).replace("void unetdc::", "")
        busy = kv["mfma_busy_frac"]             # tools/summarize_sq.py: MFMA busy cycles / (kernel-active cycles x 1024 SIMDs)
        out.append(f"| `{kk}` ({rec['op_bench_args']}) | {busy * 100:.1f} % | {cn['SQ_INSTS_LDS'] / cn['SQ_INSTS_MFMA']:.2f} | "
                   f"{cn['SQ_INSTS_VALU'] / cn['SQ_INSTS_MFMA']:.1f} | {cn['SQ_WAIT_ANY'] / cn['SQ_WAVE_CYCLES'] * 100:.0f} % |")
v["SQR3"] = "\n".join(out) + "\n"

for doc in ("DESIGN.md", "README.md"):
    s = open(os.path.join(ROOT, "tools", "templates", doc + ".in")).read()
    missing = set(re.findall(r"@@([A-Z0-9]+)@@", s)) - set(v)
    if missing:
        sys.exit(f"{doc}: no value for {sorted(missing)}")
    for k, val in v.items():
        s = s.replace(f"@@{k}@@", val)
    open(os.path.join(ROOT, doc), "w").write(s)
    print("rendered", doc)
