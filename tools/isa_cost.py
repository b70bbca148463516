"""Static instruction mix of one kernel from the compiler's assembly (hipcc -save-temps), weighted by the issue costs
measured with tools/ubench_valu.hip on MI355X (cycles per wave-instruction on one SIMD).  Diagnostic only.
  usage: python tools/isa_cost.py FILE.s KERNEL_SUBSTRING [--top N]
Straight-line count: every instruction of the kernel body once (loops of the hot kernels are fully unrolled; branches
over cold paths inflate the figure, so compare variants of the same kernel, and the PMC counters for absolute numbers)."""
import collections
import re
import sys

COST = {  # measured, 4 waves per SIMD (profiles/r03_ubench_valu.txt); default 4.5
    "v_rsq_f64": 17, "v_rcp_f64": 17, "v_sqrt_f64": 17, "v_trig_preop_f64": 17,
    "v_fma_f64": 5.8, "v_mul_f64": 5.4, "v_add_f64": 5.0, "v_mul_lo_u32": 5.5, "v_mul_hi_u32": 5.3, "v_mad_u64_u32": 5.5,
    "v_exp_f32": 8.3, "v_log_f32": 8.3, "v_sin_f32": 8.7, "v_cos_f32": 8.7, "v_sqrt_f32": 8.4, "v_rcp_f32": 8.3, "v_rsq_f32": 8.3,
    "v_xor_b32": 3.1, "v_mov_b32": 2.8, "ds_read_b64": 9.7, "ds_read_b128": 13.0,
}


def kernel_body(path, sub):
    out, on = [], False
    for line in open(path):
        if not on:
            m = re.match(r"^(\S+):", line)
            if m and sub in m.group(1) and not m.group(1).startswith("."):
                on = True
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            break
        out.append(line)
    return out


def main():
    path, sub = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    body = kernel_body(path, sub)
    ops = collections.Counter()
    for line in body:
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith(("v_", "s_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
            ops[re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op)] += 1
    valu = {k: v for k, v in ops.items() if k.startswith("v_")}
    n_valu = sum(valu.values())
    cyc = sum(COST.get(k, 4.5) * v for k, v in valu.items())
    print(f"kernel *{sub}*: {sum(ops.values())} instructions, VALU {n_valu} (cost-weighted {cyc:.0f} cycles = {cyc / 4.5:.0f} plain-op equivalents), "
          f"SALU {sum(v for k, v in ops.items() if k.startswith('s_'))}, LDS {sum(v for k, v in ops.items() if k.startswith('ds_'))}, "
          f"VMEM {sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_', 'flat_')))}")
    for k, v in sorted(ops.items(), key=lambda kv: -COST.get(kv[0], 4.5) * kv[1] if kv[0].startswith("v_") else -kv[1])[:top]:
        print(f"  {k:28s} {v:6d}  x {COST.get(k, 4.5) if k.startswith('v_') else 0:5.1f}")


if __name__ == "__main__":
    main()
