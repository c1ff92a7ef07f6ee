"""Builds libfacodec_hip.so (gfx950) in-tree with hipcc.  No torch extension glue: the library is
a plain C-ABI shared object (include/facodec_hip.h) loaded through ctypes."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# FAC_BUILD_TAG=<tag> builds a tuning variant (e.g. with FAC_EXTRA_FLAGS=-DFAC_ABL_...) next to the product library:
# libfacodec_hip_<tag>.so, objects under csrc/build_<tag>/; select it at run time with FAC_LIB_PATH.
TAG = os.environ.get("FAC_BUILD_TAG", "")
LIB = os.path.join(HERE, "libfacodec_hip%s.so" % ("_" + TAG if TAG else ""))
OBJDIR = "build" + ("_" + TAG if TAG else "")
SOURCES = ["conv1d_api.hip", "conv1d_tile_128x128.hip", "conv1d_tile_96x128.hip", "conv1d_tile_64x128.hip",
           "conv1d_tile_32x256.hip", "conv1d_tile_128x32.hip", "conv1d_tile_128x256.hip", "conv1d_fused_ru.hip", "conv1d_narrow.hip", "conv1d_pw.hip", "conv1d_pw_split.hip", "conv1d_skinny.hip", "conv1d_bsplit.hip", "conv1d_gemm_split.hip", "conv1d_bsplit2.hip", "conv1d_bwd.hip", "conv1d_wgrad_split.hip", "conv1d_wgrad_k1.hip", "train_misc.hip", "optim.hip", "train_quant.hip", "train_pred.hip", "train_disc.hip", "pack.hip", "prep_batch.hip", "lstm.hip", "lstm_persist.hip", "vq.hip", "misc.hip", "rccl_arena.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result"] + os.environ.get("FAC_EXTRA_FLAGS", "").split()


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    tt = os.path.getmtime(target)
    return any(os.path.getmtime(d) > tt for d in deps)


def build_lib(force=False, verbose=True):
    hipcc = _hipcc()
    # every header under csrc/ (common.h, conv1d_mfma.h, inflight_regs.h, ...) + the C ABI: an edit to any of them rebuilds everything
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(HERE, "..", "include", "facodec_hip.h")]
    objs, jobs = [], []
    os.makedirs(os.path.join(CSRC, OBJDIR), exist_ok=True)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB])
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
