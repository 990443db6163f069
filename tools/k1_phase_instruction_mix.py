"""Static instruction mix per profiling phase of the geometry backward's MLP kernel: compiles
hashgrid_mfma.hip with -DDSU_BWD_PROF (the s_memtime markers of tools/sdf_bwd_phase_clocks.py) for
gfx950 — no GPU needed — and counts, between consecutive markers of the <NL=10, split, feature
cache> kernel's first point half, the instructions by issue class.  Put next to the measured phase
clocks this separates issue time from stalls.  usage: python tools/k1_phase_instruction_mix.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "drawingspinup_amd", "csrc", "hashgrid_mfma.hip")
out = os.path.join(tempfile.gettempdir(), "k1_prof.s")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics",
       "-ffp-contract=off", "-Wno-unused-result", "-w", "-DDSU_BWD_PROF", "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-o", out, src]
subprocess.run(cmd, check=True)
txt = open(out).read().split("\n")
start = next(i for i, l in enumerate(txt) if l.startswith("_ZN12_GLOBAL__N_122sdf_fd_bwd_mfma_kernelILi10ELb1ELb1E"))
end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
body = txt[start:end]


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "scratch_", "buffer_")): return "vmem"
    if op.startswith("v_accvgpr"): return "accmov"
    if op.startswith(("v_readlane", "v_writelane", "v_permlane", "v_mov_b32_dpp")) or "dpp" in op: return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"


phases, cur = [], {}
for l in body:
    s = l.strip()
    if not s or s.startswith((";", ".")) or s.endswith(":"):
        continue
    op = s.split()[0]
    if op == "s_memtime":
        phases.append(cur)
        cur = {}
        continue
    c = cls(op)
    cur[c] = cur.get(c, 0) + 1
phases.append(cur)
keys = ["mfma", "valu", "trans", "lds", "vmem", "accmov", "lane", "salu", "branch", "wait", "nop"]
print("segment " + " ".join(f"{k:>6s}" for k in keys) + "   issue clocks (mfma 64, trans 16, others 4; waits excluded)")
for i, p in enumerate(phases):
    clk = 64 * p.get("mfma", 0) + 16 * p.get("trans", 0) + 4 * sum(v for k, v in p.items() if k not in ("mfma", "trans", "wait"))
    if sum(p.values()) < 4:
        continue
    print(f"{i:7d} " + " ".join(f"{p.get(k, 0):6d}" for k in keys) + f"   {clk:8d}")
