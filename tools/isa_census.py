"""Register / LDS / occupancy figures and a static instruction census of the hot kernels, from the compiler's
own output.  Builds the N = 4096 kernels only (seconds), keeps nothing but two small text files.

usage: python tools/isa_census.py [out_dir]      (default: profiles/, files r03_resource_usage.txt, r03_isa_census.txt)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = ("analysis_fast16_kernelILi11ELb1ELi4EdE", "synthesis_fast_kernelILi11ELb1ELi2EdLb0EE", "analysis_fast_kernelILi11ELb1ELi2EdE",
       "synthesis_fast16_kernelILi11ELi4EdLb0EE", "phase_carry_kernelIdE")


def classify(op):
    if op.startswith(("v_cndmask", "v_cmp", "v_cmpx")):
        return "select / compare"
    if op.endswith("_f64") or "_f64_" in op:
        return "fp64 arithmetic / convert"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap")):
        return "moves"
    if op.startswith("v_"):
        return "other vector (32-bit integer, address, bit ops)"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "global / scratch memory"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "scalar"
    return "other"


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
    src = os.path.join(ROOT, "vocoder_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-shared", "-fvisibility=hidden", "-DVOC_DEV_ONLY_LOG2M=11", "-save-temps=obj",
               "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(tmp, "voc.so"), os.path.join(src, "vocoder_hip.cpp")]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        remarks = r.stderr
        asm = next(os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f)
        text = open(asm).read()
    # ---- resource usage: the remark block of each hot kernel
    blocks = re.split(r"(?=remark: [^\n]*Function Name:)", remarks)
    with open(os.path.join(out_dir, "r03_resource_usage.txt"), "w") as f:
        f.write("# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, N = 4096 kernels of the metric configuration.\n"
                "# LDS is sized at launch (dynamic).  The kernels of the metric path: analysis_fast16 (sixteen points per thread: 32 768 B per\n"
                "# workgroup of 128 threads, 4 workgroups = 8 waves per CU) and synthesis_fast (eight points per thread: 49 168 B per workgroup\n"
                "# of 256 threads, built for 2 waves per SIMD = 2 workgroups per CU).  The other two are the alternatives they were measured against.\n")
        for b in blocks:
            if any(h in b for h in HOT):
                for line in b.splitlines():
                    if "remark:" in line:
                        line = re.sub(r"^.*remark: \S+ ", "", line).replace("[-Rpass-analysis=kernel-resource-usage]", "")
                        f.write(line.rstrip() + "\n")
                f.write("\n")
    # ---- static instruction census per hot kernel
    with open(os.path.join(out_dir, "r03_isa_census.txt"), "w") as f:
        f.write("# Static count of the instructions in the code object of each hot kernel (every instruction once,\n"
                "# whatever loop it sits in; the per-frame loop with its unrolled bin loop is most of each kernel).\n")
        for h in HOT:
            m = re.search(r"^(_ZN3voc\d+" + re.escape(h) + r"[^\n:]*):[^\n]*\n(.*?)^\s*s_endpgm", text, re.S | re.M)
            if not m:
                f.write(f"{h}: not found\n")
                continue
            # instructions in order, with the labels that precede them
            instrs, labels = [], {}
            for line in m.group(2).splitlines():
                line = line.split(";")[0].strip()
                if line.endswith(":") and line.startswith(".LBB"):
                    labels[line[:-1]] = len(instrs)
                    continue
                if not line or line.endswith(":") or line.startswith("."):
                    continue
                instrs.append(line)
            # the frame loop: the longest span closed by a backward branch
            best = (0, 0)
            for i, ins in enumerate(instrs):
                parts = ins.split()
                if parts[0].startswith(("s_cbranch", "s_branch")) and parts[-1] in labels and labels[parts[-1]] <= i:
                    if i - labels[parts[-1]] > best[1] - best[0]:
                        best = (labels[parts[-1]], i + 1)
            f.write(f"\n{m.group(1)}\n")
            for title, part in (("whole kernel", instrs), ("longest loop (one frame)", instrs[best[0]:best[1]])):
                counts, ops = collections.Counter(), collections.Counter()
                for ins in part:
                    op = ins.split()[0]
                    counts[classify(op)] += 1
                    ops[op] += 1
                total = max(sum(counts.values()), 1)
                f.write(f" {title}\n")
                for k, v in counts.most_common():
                    f.write(f"  {k:50s} {v:6d}  {100.0 * v / total:5.1f} %\n")
                f.write(f"  {'total':50s} {total:6d}\n")
                f.write("  most frequent: " + ", ".join(f"{o} {n}" for o, n in ops.most_common(16)) + "\n")
    print(open(os.path.join(out_dir, "r03_resource_usage.txt")).read())
    print(open(os.path.join(out_dir, "r03_isa_census.txt")).read())


if __name__ == "__main__":
    main()
