"""Static per-kernel statistics of the gfx950 code hipcc generates for csrc/*.hip — runs without
a GPU.  For every kernel: registers, spills, scratch, LDS (from the `amdhsa.kernels` metadata,
parsed per kernel block) and instruction counts that have pointed at real problems before
("pitfalls"): v_accvgpr_mov per MFMA (accumulators copied around conditional MFMAs),
exec-mask branch regions (conditional loads), scratch traffic, v_readlane/v_writelane (SGPR spills),
`v_max_f32 v, v, v` (fmaxf quieting a possible sNaN: a second VALU slot per ReLU).

    python tools/isa_stats.py [file.hip ...] [--filter substring] [--loops]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "drawingspinup_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffp-contract=off",
         "-S", "--cuda-device-only"]


def compile_asm(src):
    out = os.path.join(tempfile.gettempdir(), "isa_" + os.path.basename(src) + ".s")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run([hipcc, *FLAGS, "-o", out, os.path.abspath(src)], capture_output=True, text=True,
                       cwd=tempfile.gettempdir())
    if r.returncode != 0:
        raise SystemExit(r.stderr)
    return open(out).read()


def metadata(txt):
    md = txt[txt.index("amdhsa.kernels:"):]
    out = {}
    for block in md.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        def g(k):
            m = re.search(r"\." + k + r":\s+(\d+)", block)
            return int(m.group(1)) if m else 0
        out[name] = dict(agpr=int(block.split("\n")[0].strip()), vgpr=g("vgpr_count"),
                         sgpr=g("sgpr_count"), vspill=g("vgpr_spill_count"),
                         sspill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"),
                         lds=g("group_segment_fixed_size"))
    return out


def bodies(txt):
    lines = txt.split("\n")
    cur, buf = None, []
    for ln in lines:
        m = re.match(r"^(_Z\S+):\s", ln)
        if m and cur is None:
            cur, buf = m.group(1), []
            continue
        if cur is not None:
            if "s_endpgm" in ln:
                yield cur, buf
                cur = None
            else:
                buf.append(ln)


def demangle_short(name):
    name = re.sub(r"^_ZN\d+_GLOBAL__N_1", "", name)
    name = re.sub(r"^_Z", "", name)
    m = re.match(r"(\d+)", name)
    if m:
        n = int(m.group(1))
        rest = name[len(m.group(1)):]
        targs = re.findall(r"L[ib](\d+)E", rest[n:n + 60].split("EEv")[0]) if rest[n:n + 1] == "I" else []
        return rest[:n] + ("<" + ",".join(targs) + ">" if targs else "")
    return name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--filter", default="")
    ap.add_argument("--loops", action="store_true", help="instruction count per loop nest level")
    a = ap.parse_args()
    files = a.files or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    print(f"{'kernel':46s} {'vgpr':>4s} {'agpr':>4s} {'vsp':>3s} {'ssp':>3s} {'scr':>4s} {'lds':>6s} "
          f"{'instr':>6s} {'mfma':>4s} {'accmov':>6s} {'brreg':>5s} {'scrop':>5s} {'lane':>4s} {'nop':>4s} {'quiet':>5s}")
    for f in files:
        txt = compile_asm(f)
        md = metadata(txt)
        for name, body in bodies(txt):
            short = demangle_short(name)
            if a.filter and a.filter not in short:
                continue
            c = Counter()
            depth = Counter()
            cur_depth = 0
            for ln in body:
                m = re.search(r"Depth=(\d+)", ln)
                if ln.startswith(".LBB") or ln.startswith("; %bb."):
                    cur_depth = int(m.group(1)) if m else 0
                    continue
                mi = re.match(r"\s+([a-z_0-9]+)", ln)
                if mi:
                    c[mi.group(1)] += 1
                    depth[cur_depth] += 1
                    # v_max_f32 v, v, v: fmaxf()'s quieting of a possible sNaN ahead of the real max
                    if re.match(r"\s+v_max_f32_e32 (v\d+), (v\d+), \2\s*$", ln):
                        c["_quiet"] += 1
            m = md.get(name, {})
            mfma = sum(v for k, v in c.items() if k.startswith("v_mfma"))
            scr = sum(v for k, v in c.items() if k.startswith("scratch_"))
            lane = c["v_readlane_b32"] + c["v_writelane_b32"]
            print(f"{short[:46]:46s} {m.get('vgpr', 0):4d} {m.get('agpr', 0):4d} {m.get('vspill', 0):3d} "
                  f"{m.get('sspill', 0):3d} {m.get('scratch', 0):4d} {m.get('lds', 0):6d} "
                  f"{sum(c.values()) - c['_quiet']:6d} {mfma:4d} {c['v_accvgpr_mov_b32']:6d} "
                  f"{c['s_cbranch_execz']:5d} {scr:5d} {lane:4d} {c['s_nop']:4d} {c['_quiet']:5d}")
            if a.loops:
                print("      instructions by loop depth:", dict(sorted(depth.items())))


if __name__ == "__main__":
    sys.exit(main())
