"""Build the gfx950 shared library (C ABI in include/lfvdm_hip.h) with hipcc.

    python latent-flexible-video-diffusion-modeling_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU; the resulting ``lib/liblfvdm_hip.so`` is
git-ignored but travels to the GPU box with the gpurun snapshot.

What decides a recompile is CONTENT, not mtimes: ``lib/build_manifest.json`` (next to the library) records, per
object, the sha256 of its source + every header + the compile flags, and for the library the hashes of the objects it
was linked from.  An object is reused only if its recorded input hash equals the current one AND the object file on
disk still has the recorded digest; a stale or foreign ``.o`` is therefore never linked silently.
``build_report()`` returns what the last ``build()`` did: ``{"lib", "compiled": [...], "reused": [...], "linked"}``.
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liblfvdm_hip.so")
MANIFEST = os.path.join(LIBDIR, "build_manifest.json")
ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17"]

_last_report = None


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def header_files():
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")))


def source_files():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def input_hash(src, header_digests, extra=()):
    """Digest of everything that determines the object of `src`: its text, every header, the flags."""
    h = hashlib.sha256()
    h.update(_sha(src).encode())
    for name, dig in header_digests:
        h.update(name.encode())
        h.update(dig.encode())
    h.update(" ".join(FLAGS + list(extra)).encode())
    return h.hexdigest()


def _load_manifest(path=None):
    try:
        with open(path or MANIFEST) as f:
            m = json.load(f)
        return m if isinstance(m, dict) else {}
    except (OSError, ValueError):
        return {}


def build(force=False, verbose=True, header_salt="", libdir=None):
    """Compile what changed, link, record.  `header_salt` is folded into the header digest (the build test uses it to
    stand for "a header's content hash changed" without editing a file); `libdir` builds somewhere else than lib/."""
    global _last_report
    if libdir is not None:
        return _build_in(libdir, force, verbose, header_salt)
    return _build_in(LIBDIR, force, verbose, header_salt)


def _build_in(LIBDIR, force, verbose, header_salt):
    global _last_report
    LIB = os.path.join(LIBDIR, "liblfvdm_hip.so")
    MANIFEST = os.path.join(LIBDIR, "build_manifest.json")
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr_digests = [(os.path.relpath(h, ROOT), _sha(h)) for h in header_files()]
    if header_salt:
        hdr_digests.append(("<salt>", header_salt))
    old = _load_manifest(MANIFEST).get("objects", {})
    objs, procs, compiled, reused, hashes = [], [], [], [], {}
    for s in source_files():
        name = os.path.basename(s)[:-4]
        o = os.path.join(LIBDIR, name + ".o")
        objs.append(o)
        want = input_hash(s, hdr_digests)
        hashes[name] = want
        rec = old.get(name)
        fresh = (not force and rec is not None and rec.get("inputs") == want and os.path.exists(o)
                 and _sha(o) == rec.get("object"))
        if fresh:
            reused.append(name)
            continue
        cmd = [hipcc] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", s, "-o", o]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        compiled.append(name)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    objects = {os.path.basename(o)[:-2]: {"inputs": hashes[os.path.basename(o)[:-2]], "object": _sha(o)} for o in objs}
    link_inputs = hashlib.sha256(json.dumps(sorted((k, v["object"]) for k, v in objects.items())).encode()).hexdigest()
    oldm = _load_manifest(MANIFEST)
    linked = bool(force or procs or not os.path.exists(LIB) or oldm.get("link_inputs") != link_inputs
                  or oldm.get("lib") != _sha(LIB))
    if linked:
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(MANIFEST, "w") as f:
        json.dump({"arch": ARCH, "flags": FLAGS, "objects": objects, "link_inputs": link_inputs, "lib": _sha(LIB),
                   "source_digest": source_digest()}, f, indent=1, sort_keys=True)
    _last_report = {"lib": LIB, "compiled": compiled, "reused": reused, "linked": linked}
    return LIB


def source_digest():
    """One digest over csrc/* + headers: what profiles are stamped with (bench.py flags records taken from other code)."""
    h = hashlib.sha256()
    for p in source_files() + header_files():
        h.update(os.path.relpath(p, ROOT).encode())
        h.update(_sha(p).encode())
    return h.hexdigest()[:16]


def build_report():
    return dict(_last_report) if _last_report else None


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(json.dumps(build_report()))
