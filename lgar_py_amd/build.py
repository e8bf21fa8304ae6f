"""Build the HIP library (liblgar_hip.so) in-tree for gfx950 with hipcc."""
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
           "lgar_tangent_body.hpp", os.path.join("..", "..", "include", "lgar.h")]
# -ffp-contract=off: expression rounding follows the reference's Python (no FMA contraction)
# fp32 division stays correctly rounded: with the rcp-based fast divide x/x != 1, Se = (theta-theta_r)/(theta_e-theta_r)
# exceeds 1 at saturation and 8 % of perturbed columns fault (measured), for no speed gain.
# -munsafe-fp-atomics: atomicAdd(double*) is one global_atomic_add_f64, not a compare-and-swap loop
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-munsafe-fp-atomics", "-fPIC", "-shared", "-std=c++17", "-Wno-pass-failed"]
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


def _stamp(lib, extra=()):
    with open(lib + ".sha256", "w") as fh:
        fh.write(_fingerprint(extra) + "\n")


def _compile_all(units, objdir, tag, extra, verbose):
    """Compile translation units concurrently (JOBS at a time); an object is reused if it is newer than every source."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build liblgar_hip.so")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra)
    newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in SOURCES + HEADERS if os.path.exists(os.path.join(CSRC, f)))
    objs, todo = [], []
    for src, uflags, suffix in units:
        obj = os.path.join(objdir, "%s%s%s.o" % (src[:-4], suffix, tag))
        objs.append(obj)
        if not (os.path.exists(obj) and os.path.getmtime(obj) >= newest):
            todo.append([hipcc] + cflags + uflags + ["-c", os.path.join(CSRC, src), "-o", obj])
    running = []
    while todo or running:
        while todo and len(running) < JOBS:
            cmd = todo.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
        cmd, p = running.pop(0)
        if p.wait() != 0:
            for _, q in running:
                q.kill()
            raise subprocess.CalledProcessError(p.returncode, cmd)
    return hipcc, objs


def build_variant(name, extra_flags, verbose=False, layers=(3,), tangent=False):
    """Measurement variants (tools/ablate.py): the same sources with extra -D flags -> csrc/variants/liblgar_hip_<name>.so
    (select one at run time with LGAR_LIB=<path>); forward path of the given layer counts only unless tangent=True."""
    vdir = os.path.join(CSRC, "variants")
    out = os.path.join(vdir, "liblgar_hip_%s.so" % name)
    if not _stale(out, extra_flags):
        return out
    units = [u for u in UNITS if not u[2] or int(u[2][1:]) in layers]
    if not tangent:
        units = [u for u in units if u[0] != "lgar_tangent_nl.hip"]
    flags = list(extra_flags) + ["-DLGAR_ONLY_LAYERS=%s" % "".join(str(n) for n in layers)] + ([] if tangent else ["-DLGAR_NO_TANGENT"])
    hipcc, objs = _compile_all(units, os.path.join(vdir, "obj"), "_" + name, flags, verbose)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    _stamp(out, extra_flags)
    return out


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> csrc/liblgar_hip.so.  hipcc cross-compiles gfx950 without a GPU.  The translation units
    (C-ABI, probe, and one per soil-layer count for the forward and the tangent kernels) are compiled concurrently into
    csrc/obj/ and then linked."""
    if not force and not _stale():
        return LIB
    objdir = os.path.join(CSRC, "obj")
    if force and os.path.isdir(objdir):
        shutil.rmtree(objdir)
    hipcc, objs = _compile_all(UNITS, objdir, "", [], verbose)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    _stamp(LIB)
    return LIB
