"""LDS traffic per kernel from the SQ_INSTS_LDS pass (tools/pmc_summary.py output of pass p2):
{"kernels": {"<kernel> grid=<n>": {"SQ_INSTS_LDS": wave-level LDS instructions per launch, "bytes_per_lane": the mean access
width the kernel's source uses, "lds_bytes_per_launch": insts * 64 lanes * bytes_per_lane, "conflict_ratio":
SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS}}}.  The counters give instruction counts, not bytes: the width per lane is
read off the source of each kernel (listed below) and is the one assumption in the resulting GB/s."""
import json, re, sys

# mean bytes per lane and LDS instruction, from the kernels' sources
WIDTH = {
    "blend_forward_kernel": (13.0, "per visit two ds_read_b128 (s_a, s_b) + one ds_read_b64 (s_c); staging writes of the same widths"),
    "blend_backward_kernel": (12.0, "per visit ds_read_b128 / b64 of the staged entry, b32 row stores of the per-wave reduction"),
    "segment_depth_sort_kernel": (6.0, "64-bit key / 32-bit value moves and 32-bit counter read-modify-writes"),
    "sort_scatter_kernel": (6.0, "64-bit keys + 32-bit values through the LDS exchange buffer"),
    "sort_hist_kernel": (4.0, "32-bit histogram atomics"),
    "emit_keys_kernel": (4.0, "32-bit scan / search words"),
    "heads_forward_kernel": (16.0, "ds_read_b128 weight operands"),
    "heads_backward_kernel": (16.0, "ds_read_b128 weight operands"),
    "dw_partial_kernel": (12.0, "float4 staging stores, b32 / b64 operand reads"),
    "pe_forward_kernel": (8.0, "column assembly, float4 row reads"),
}
out = {}
for line in open(sys.argv[1]):
    m = re.match(r"(.*) grid=(\d+) lds=\d+ (.*)", line.strip())
    if not m:
        continue
    c = dict(kv.split("=") for kv in m.group(3).split())
    insts = float(c.get("SQ_INSTS_LDS", 0))
    if insts <= 0:
        continue
    name = m.group(1)
    width, why = next(((w, y) for k, (w, y) in WIDTH.items() if name.startswith(k)), (8.0, "default: 8 bytes per lane"))
    act = float(c.get("SQ_ACTIVE_INST_LDS", 0))
    out[f"{name} grid={m.group(2)}"] = {"SQ_INSTS_LDS": insts, "bytes_per_lane": width, "width_source": why,
                                        "lds_bytes_per_launch": insts * 64 * width,
                                        "conflict_ratio": round(float(c.get("SQ_LDS_BANK_CONFLICT", 0)) / act, 3) if act else None}
print(json.dumps({"source": "rocprofv3 --pmc pass p2 (SQ_INSTS_LDS, SQ_ACTIVE_INST_LDS, SQ_LDS_BANK_CONFLICT) of tools/profile_round.sh; "
                            "bytes = instructions x 64 lanes x bytes_per_lane (width read off each kernel's source)",
                  "kernels": out}, indent=1))
