"""Builds ssme_amd/libssme_pf.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libssme_pf.so")
SOURCES = [os.path.join(CSRC, "pf_api.hip")]
STAMP = SO + ".srchash"          # hash of every source + the flags the .so was built from (ships with the .so)


def deps():
    """Everything the library is compiled from: csrc/* and the C-ABI header."""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))) + [
        os.path.join(os.path.dirname(HERE), "include", "ssme_pf.h")]

# -ffp-contract=off: the libm-free math is a fixed IEEE operation sequence (explicit fma only)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wl,-soname,libssme_pf.so", "-ffp-contract=off", "-mfma",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def source_hash(extra=()):
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + list(extra)).encode())
    for d in deps():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    """True unless the .so on disk was built from exactly these sources and flags (content hash, not mtimes:
    a snapshot copied to another box keeps no useful timestamps)."""
    if not os.path.exists(SO) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=False, extra=(), out=None):
    """out: alternative output path (measurement builds, e.g. extra=("-DSSME_ABLATE",))."""
    if out is None and not force and not needs_build():
        return SO
    if out is None:
        # Every rank of a torchrun launch calls lib() -> build(): one of them compiles, the others wait on the lock and then
        # find the stamp current.  The compiler writes to a temporary file beside the target; the resource checks run on that
        # build, and only then is the .so replaced (atomically) and the stamp written -- no rank can dlopen a half-written
        # library or pass needs_build() between the two writes.
        import fcntl
        with open(SO + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if not force and not needs_build():
                    return SO
                tmp = SO + ".tmp%d" % os.getpid()
                try:
                    _compile(extra, tmp, verbose)
                    os.replace(tmp, SO)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
                if not extra:
                    with open(STAMP + ".tmp", "w") as f:
                        f.write(source_hash() + "\n")
                    os.replace(STAMP + ".tmp", STAMP)
                return SO
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    _compile(extra, out, verbose)
    return out


def _compile(extra, target, verbose):
    cmd = [hipcc()] + FLAGS + list(extra) + SOURCES + ["-o", target]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise subprocess.CalledProcessError(res.returncode, cmd)
    # No kernel may keep data in scratch memory: a spilled array silently doubles the HBM traffic of the step kernel.
    # Allowed: a frame of <= 128 bytes in a kernel that spills SGPRs only (> 102 live uniform values; the compiler
    # parks them in VGPR lanes with v_writelane/v_readlane and reserves a small frame it does not address).
    stats, name = {}, None
    for line in res.stdout.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
            stats[name] = {}
        for key in ("ScratchSize [bytes/lane]", "SGPRs Spill", "VGPRs Spill"):
            if name and key + ":" in line:
                stats[name][key] = int(line.split(key + ":")[1].split()[0])
    for name, st in stats.items():
        if "k_test" in name:
            continue
        if st.get("VGPRs Spill", 0) != 0:
            raise RuntimeError(f"kernel {name} spills vector registers: {st}")
        scratch = st.get("ScratchSize [bytes/lane]", 0)
        if scratch != 0 and not (st.get("SGPRs Spill", 0) > 0 and scratch <= 128):
            raise RuntimeError(f"kernel {name} uses scratch memory: {st}")


def build_user_model(header, name, verbose=False):
    """The library with ONE user model compiled in as SSME_MODEL_USER0 (ssme_amd/csrc/model_api.h): header = a C++ header
    defining `struct ssme_user_model0`.  Returns build/user/libssme_pf_<name>.so; use it with SSME_PF_LIB=<that path>."""
    header = os.path.abspath(header)
    if not os.path.exists(header):
        raise FileNotFoundError(header)
    out_dir = os.path.join(os.path.dirname(HERE), "build", "user")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"libssme_pf_{name}.so")
    import hashlib
    with open(header, "rb") as f:
        tag = hashlib.sha256(f.read() + source_hash().encode() + b"soname=file").hexdigest()
    stamp = out + ".srchash"
    if os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == tag:
        return out
    # its own soname (the last -soname wins): a program linked against this file then looks for THIS file, not for libssme_pf.so
    build(force=True, verbose=verbose, extra=(f'-DSSME_USER_MODEL_HEADER="{header}"', f"-Wl,-soname,{os.path.basename(out)}"), out=out)
    with open(stamp, "w") as f:
        f.write(tag + "\n")
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
