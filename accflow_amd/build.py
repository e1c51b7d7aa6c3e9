"""Build libaccflow_hip.so (gfx950) in-tree with hipcc.

The library is linked WITHOUT a HIP runtime dependency (-no-hip-rt): its hip* symbols resolve at load
time against the HIP runtime the host process already uses (PyTorch-ROCm's bundled libamdhip64 when
loaded through accflow_amd._lib, /opt/rocm's for a plain C/C++ host).  Two HIP runtimes in one process
would not share streams or allocations, which is why the library must not pull in its own.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libaccflow_hip.so")
# (heaviest translation units first: the pool runs 8 at a time; the fp32-MFMA kernel's 14 instantiations are spread over
# four units since round 5 - as ONE unit they took ~6.5 minutes and bounded a from-scratch build)
SOURCES = ["conv2d_f32.hip", "conv2d_f32_v1.hip", "conv2d_f32_v2.hip", "conv2d_f32_v3.hip", "conv2d_bf16s.hip", "conv2d_bf16s_v12.hip", "conv2d_bf16s_v11.hip", "conv2d_bf16s_v21.hip",
           "conv2d_direct_v_bf16x6.hip", "conv2d_direct_v_bf16n.hip", "conv2d_direct_v_f16.hip", "conv2d_direct_v_s16.hip", "conv2d_direct_v_s16tg.hip", "conv2d_direct_v_s16k.hip", "conv2d_direct_v_s16k9.hip",
           "conv2d_direct_v_f16n.hip", "conv2d_direct_v_bf16x3.hip", "conv_s16m_v0.hip", "conv_s16m_v1.hip", "conv_s16m_v2.hip",
           "conv_s16m_v3.hip", "conv2d_s16m.hip", "conv_stem.hip", "conv2d_direct.hip", "conv2d.hip", "corr_volume.hip",
           "corr_lookup.hip", "corr_disp.hip", "corr_lookup_conv.hip", "sampling.hip", "misc.hip", "gma.hip", "backward.hip"]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _deps(src, seen=None):
    """The source and every header it includes (transitively) from csrc/ or include/: a unit is rebuilt when one of THESE
    changed, not when any header did (the kernel templates live in headers; a full rebuild takes ~8 minutes)."""
    import re
    seen = set() if seen is None else seen
    if src in seen:
        return seen
    seen.add(src)
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(src).read(), flags=re.M):
        for d in (os.path.dirname(src), CSRC, os.path.join(ROOT, "include")):
            cand = os.path.join(d, name)
            if os.path.exists(cand):
                _deps(cand, seen)
                break
    return seen


def build(force=False, verbose=False, libdir=None, defines=(), unit_defines=()):
    """libdir / defines / unit_defines: experiment builds (tools/): another output directory, extra -D flags for every unit,
    and (source-name prefix, define) pairs for -D flags of single units; the product build uses none of them.  Select such
    a library at run time with ACCFLOW_HIP_LIB=<libdir>/libaccflow_hip.so.  (An experiment directory seeded with a copy of
    lib/obj - cp -a, the time stamps matter - recompiles only the units whose flags differ: tools/precision_probe_s16m.sh.)"""
    LIBDIR = libdir or globals()["LIBDIR"]
    LIB = os.path.join(LIBDIR, "libaccflow_hip.so")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    cc = _hipcc()
    flags = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
             "-I" + CSRC, "-Wno-unused-result",
             # fully unroll the (large) epilogue loops so that 96-128-register accumulator arrays stay in VGPRs
             "-mllvm", "-pragma-unroll-threshold=1000000"] + ["-D" + d for d in defines]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        extra = ["-D" + d for pre, d in unit_defines if s.startswith(pre)]
        if force or extra or not _newer(obj, sorted(_deps(src))):
            jobs.append([cc] + flags + extra + ["-c", src, "-o", obj])

    def run(cmd):
        import time
        if verbose:
            print(" ".join(cmd), flush=True)
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose:
            print("  [%.0f s] %s" % (time.time() - t0, os.path.basename(cmd[-1])), flush=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-no-hip-rt", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    ld = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--libdir=")]
    ud = [tuple(a.split("=", 1)[1].split(":", 1)) for a in sys.argv[1:] if a.startswith("--unit-define=")]
    print(build(force="--force" in sys.argv, verbose=True, libdir=ld[0] if ld else None, defines=defs, unit_defines=ud))
