"""Group a rocprofv3 kernel_stats.csv by kernel class and print ms per step.  Usage: class_sums.py stats.csv steps"""
import csv, sys, re
path, steps = sys.argv[1], float(sys.argv[2])
classes = [("igemm fwd/dgrad (conv + Linear)", r"igemm_kernel|conv3x3_c64|conv8p_kernel|gemm8p_kernel"), ("wgrad (all weight gradients + reduce)", r"wgrad|slab"),
           ("stem conv / pool fwd", r"stem_conv_kernel|stem_pool_fwd|stem_pack"), ("BatchNorm passes", r"bn_|stem_bwd_reduce|stem_bwd_apply"),
           ("SE / spatial", r"se_|spatial_|scale_kernel"), ("attention", r"attn_"), ("LayerNorm", r"layernorm"),
           ("token elementwise (bias/act, pools, gate, embed, add, colsum)", r"bias_act|masked_pool|gate_|embed_|add_kernel|colsum|fold_rows"),
           ("loss / optimizer / casts / packs", r"cross_entropy|adamw|sumsq|convert_kernel|pack_|fold_bn|accuracy"), ("torch / copies", r".")]
tot = {c: 0.0 for c, _ in classes}
rows = list(csv.DictReader(open(path)))
for r in rows:
    for c, pat in classes:
        if re.search(pat, r["Name"]):
            tot[c] += float(r["TotalDurationNs"]); break
s = 0.0
for c, _ in classes:
    print(f"{c:70s} {tot[c] / steps / 1e6:8.3f} ms/step"); s += tot[c]
print(f"{'SUM':70s} {s / steps / 1e6:8.3f} ms/step")
