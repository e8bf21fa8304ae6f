"""Build the HIP library (liblgar_hip.so) in-tree for gfx950 with hipcc."""
import contextlib
import fcntl
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.environ.get("LGAR_LIB") or os.path.join(CSRC, "liblgar_hip.so")
LAYERS = (2, 3, 4, 5, 6)  # LGAR_LMIN .. LGAR_LMAX: one translation unit per soil-layer count and kernel family
# (source, extra flags, object suffix)
UNITS = [("lgar_kernels.hip", [], ""), ("lgar_probe.hip", [], "")] + \
        [("lgar_kernels_nl.hip", ["-DLGAR_NL=%d" % n], "_%d" % n) for n in LAYERS] + \
        [("lgar_tangent_nl.hip", ["-DLGAR_NL=%d" % n], "_%d" % n) for n in LAYERS]
SOURCES = sorted(set(u[0] for u in UNITS))
HEADERS = ["lgar_device.hpp", "lgar_dual.hpp", "lgar_math.hpp", "lgar_host.hpp", "lgar_launch.hpp", "lgar_forward_body.hpp",
           "lgar_tangent_body.hpp", "lgar_measure.hpp", os.path.join("..", "..", "include", "lgar.h")]
# -ffp-contract=off: expression rounding follows the reference's Python (no FMA contraction)
# fp32 division stays correctly rounded: with the rcp-based fast divide x/x != 1, Se = (theta-theta_r)/(theta_e-theta_r)
# exceeds 1 at saturation and 8 % of perturbed columns fault (measured), for no speed gain.
# -munsafe-fp-atomics: atomicAdd(double*) is one global_atomic_add_f64, not a compare-and-swap loop
# -Os, not -O3: the kernels are 10-25 thousand instructions each and eight waves of a CU sit at eight different places in them;
# the size-minded build is 3-4 % faster in fp32, 3 % in the mixed mode, 1-2 % elsewhere (same-box A/B of the whole library,
# profiles/r04/mixed_kernel_experiments.jsonl) -- it unrolls and duplicates less, and spills fewer registers (fp32: 31 vs 37)
FLAGS = ["--offload-arch=gfx950", "-Os", "-ffp-contract=off", "-munsafe-fp-atomics", "-fPIC", "-shared", "-std=c++17", "-Wno-pass-failed"]
JOBS = int(os.environ.get("LGAR_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 4)


def _fingerprint(extra=()):
    """sha256 over the sources, headers and flags: what a built library is checked against (file times do not survive
    every copy of the tree, contents do)."""
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(" ".join(FLAGS + list(extra)).encode())
    return h.hexdigest()


def _stale(lib=None, extra=()):
    lib = lib or LIB
    stamp = lib + ".sha256"
    if not (os.path.exists(lib) and os.path.exists(stamp)):
        return True
    with open(stamp) as fh:
        return fh.read().strip() != _fingerprint(extra)


def _stamp(lib, fingerprint):
    """Record what the library was built FROM: the fingerprint taken before the first compile (a source edited while the
    compilers ran must leave a stale stamp, not a fresh one)."""
    with open(lib + ".sha256", "w") as fh:
        fh.write(fingerprint + "\n")


@contextlib.contextmanager
def _locked(path):
    """One builder at a time per output: every rank of `bench.py --gpus N` / torchrun reaches _capi.load() -> build() together
    on a fresh checkout (the .so and its stamp are not in the repository)."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path + ".lock", "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def _compile_all(units, objroot, tag, extra, verbose):
    """Compile translation units concurrently (JOBS at a time).  Objects live in a directory named after the content
    fingerprint (sources, headers AND flags), so an object is only ever reused by the exact build it was made for: a change of
    FLAGS, of a unit's flags or of a header can never relink stale objects.  Older fingerprints' directories are removed."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build liblgar_hip.so")
    fp = hashlib.sha256((_fingerprint(extra) + "|" + repr([(u[0], u[1], u[2]) for u in UNITS])).encode()).hexdigest()[:16]
    objdir = os.path.join(objroot, fp)
    if os.path.isdir(objroot):
        for d in os.listdir(objroot):
            full = os.path.join(objroot, d)
            if d != fp and (os.path.isdir(full) and len(d) == 16 or d.endswith(".o")):
                shutil.rmtree(full, ignore_errors=True) if os.path.isdir(full) else os.remove(full)
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra)
    objs, todo = [], []
    for src, uflags, suffix in units:
        obj = os.path.join(objdir, "%s%s%s.o" % (src[:-4], suffix, tag))
        objs.append(obj)
        if not os.path.exists(obj):
            # compile to a temporary name: an interrupted compile leaves no object that looks finished
            todo.append((obj, [hipcc] + cflags + uflags + ["-c", os.path.join(CSRC, src), "-o", obj + ".part"]))
    running = []
    while todo or running:
        while todo and len(running) < JOBS:
            obj, cmd = todo.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((obj, cmd, subprocess.Popen(cmd)))
        obj, cmd, p = running.pop(0)
        if p.wait() != 0:
            for _, _, q in running:
                q.kill()
            raise subprocess.CalledProcessError(p.returncode, cmd)
        os.replace(obj + ".part", obj)
    return hipcc, objs


def _link(hipcc, objs, out, verbose=False):
    """Link to a temporary file and rename it into place: a concurrent CDLL never sees a half-written library."""
    tmp = "%s.%d.tmp" % (out, os.getpid())
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    os.replace(tmp, out)


def build_variant(name, extra_flags, verbose=False, layers=(3,), tangent=False):
    """Measurement variants (tools/ablate.py): the same sources with extra -D flags -> csrc/variants/liblgar_hip_<name>.so
    (select one at run time with LGAR_LIB=<path>); forward path of the given layer counts only unless tangent=True."""
    if len(layers) > 1 and any(f.startswith(("-DLGAR_CLOCKS", "-DLGAR_COUNT_")) for f in extra_flags):
        # lgar_debug_counters / lgar_debug_clocks (lgar_kernels_nl.hip) are defined -- with their static counter arrays -- once
        # per layer count's translation unit: two of them would collide at link time or read back the wrong unit's counters
        raise ValueError("measurement read-backs (LGAR_CLOCKS / LGAR_COUNT_*) exist per layer count: build one layer count")
    vdir = os.path.join(CSRC, "variants")
    out = os.path.join(vdir, "liblgar_hip_%s.so" % name)
    units = [u for u in UNITS if not u[2] or int(u[2][1:]) in layers]
    if not tangent:
        units = [u for u in units if u[0] != "lgar_tangent_nl.hip"]
    # -DLGAR_MEASURE: the measurement points of the device code take their definitions from csrc/lgar_measure.hpp
    flags = list(extra_flags) + ["-DLGAR_MEASURE", "-DLGAR_ONLY_LAYERS=%s" % "".join(str(n) for n in layers)] + \
            ([] if tangent else ["-DLGAR_NO_TANGENT"])
    # the stamp covers everything that decides what is in the library: the layer counts and the tangent kernels too (the same
    # name asked for again with other layers must be rebuilt, not returned as it is)
    if not _stale(out, flags):
        return out
    with _locked(out):
        if not _stale(out, flags):
            return out
        fp = _fingerprint(flags)
        hipcc, objs = _compile_all(units, os.path.join(vdir, "obj_" + name), "", flags, verbose)
        _link(hipcc, objs, out, verbose)
        _stamp(out, fp)
    return out


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> csrc/liblgar_hip.so.  hipcc cross-compiles gfx950 without a GPU.  The translation units
    (C-ABI, probe, and one per soil-layer count for the forward and the tangent kernels) are compiled concurrently into
    csrc/obj/<fingerprint>/ and then linked.  Safe to call from several processes at once (file lock; the library is renamed
    into place)."""
    if not force and not _stale():
        return LIB
    with _locked(LIB):
        if not force and not _stale():  # another process built it while this one waited for the lock
            return LIB
        objroot = os.path.join(CSRC, "obj")
        if force and os.path.isdir(objroot):
            shutil.rmtree(objroot)
        fp = _fingerprint()
        hipcc, objs = _compile_all(UNITS, objroot, "", [], verbose)
        _link(hipcc, objs, LIB, verbose)
        _stamp(LIB, fp)
    return LIB
