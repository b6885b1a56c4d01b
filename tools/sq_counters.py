"""Summarise two rocprofv3 --pmc passes over tools/bench_kernels.py (SQ counters) into per-kernel utilisation figures.
usage: python tools/sq_counters.py gpurun_out/sq1 gpurun_out/sq2 profiles/r01_sq_counters.json
pass 1: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass 2: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES
MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 (one count per shader engine)."""
import collections
import csv
import glob
import json
import sys


def load(d):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


a, b = load(sys.argv[1]), load(sys.argv[2])
out = {}
for k in a:
    if not any(t in k for t in ("igemm_kernel", "wgrad_kernel", "wgrad3x3_c", "stem_wgrad", "conv8p_kernel", "conv3x3_c64p", "wgrad_dma_kernel")) or k not in b:
        continue
    c, d = a[k], b[k]
    cyc = c["SQ_BUSY_CYCLES"] / 32.0
    out[k] = {
        "mfma_pipe_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 3),
        "wave_life_parked_waitcnt_barrier": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
        "wave_life_issue_stalled": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
        "wave_life_issuing": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
        "lds_bank_conflict_cycles_per_lds_cycle": round(d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1.0), 4),
        "insts_per_wave": {n: round(d[f"SQ_INSTS_{n}"] / max(d["SQ_WAVES"], 1.0), 1) for n in ("MFMA", "VALU", "SALU", "LDS", "VMEM")},
    }
json.dump({"note": "averages over the launches of `python tools/bench_kernels.py --iters 2` (B=512 layer shapes, bf16); "
                   "SQ_INSTS_VALU includes the MFMA instructions", "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(k[:70], v["mfma_pipe_busy"], v["wave_life_parked_waitcnt_barrier"], v["wave_life_issue_stalled"], v["lds_bank_conflict_cycles_per_lds_cycle"], v["insts_per_wave"])
