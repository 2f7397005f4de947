"""Build the gfx950 HIP extension in-tree:  openibl_amd/libopenibl_amd.so.

Plain hipcc (no cmake, no torch cpp_extension): each csrc/*.hip is compiled for gfx950 and the
objects are linked into one C-ABI shared library that `openibl_amd.lib` loads with ctypes.
hipcc cross-compiles without a GPU, so this runs in the build container; the .so is git-ignored
but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
INCLUDE = ROOT / "include"
BUILD_DIR = ROOT / "build" / "obj"
BUILD_DIR_DBG = ROOT / "build" / "obj_dbg"
LIB_PATH = PKG_DIR / "libopenibl_amd.so"          # the product: what include/openibl_amd.h declares, nothing else
LIB_PATH_DBG = PKG_DIR / "libopenibl_amd_dbg.so"  # same sources + -DOIBL_DEBUG_HOOKS: test hooks, experiment kernels
ARCH = "gfx950"

CXXFLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fno-gpu-rdc",
    "-Wall",
    "-Wno-unused-function",
    f"-I{INCLUDE}",
    f"-I{CSRC}",
]


# compile-time experiments ride in the DEBUG library only (a script times the same layer through the product
# library and through this one: tests/gpu_dbgvariant_ab.py).  Round 4: ["-DOIBL_MX_TAIL_B128"] — the f16mx fragment
# tail as one ds_read_b128 instead of b64 + b32 (same bits, 6.950 -> 6.936 ms over the layers: nothing; off again);
# ["-DOIBL_RING_LGKM0"] — lgkmcnt(0) in front of every COMPUTE segment, the schedule before the counted waits.
# Round 6: ["-DOIBL_STEM_R6_LDS"] — the f16mx stem with MX tails as one ds_read_b128 and conflict-free producer line
# writes (2612 -> 452 modelled bank-conflict cycles per tile, 2.0e8 -> 3.5e7 measured per launch): not faster (1.609
# against 1.598 ms — the kernel is bound by VALU issue: profiles/r06_d_stem_lds_ab.txt); not adopted, off again.
# ["-DOIBL_STEM_SPLIT"] — the f16mx stem of rounds 3-5 (two workgroups per tile, resident weights) as the debug library:
# tests/gpu_stem_lds_ab.py times it against the product (one workgroup per tile): 1.555 -> 1.384 ms, bit-identical,
# profiles/r06_j_stem_dual_ab.txt — adopted; off again.  ["-DOIBL_HALO4_CONT"]: profiles/r06_e_halo4_cont.txt.
DBG_EXPERIMENT_FLAGS = []


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: the MI355X extension cannot be built on this machine")
    return exe


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _digest() -> str:
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*")) + [INCLUDE / "openibl_amd.h", Path(__file__)]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(CXXFLAGS).encode())
    return h.hexdigest()


def is_current() -> bool:
    stamp = BUILD_DIR / "stamp"
    return LIB_PATH.exists() and LIB_PATH_DBG.exists() and stamp.exists() and stamp.read_text() == _digest()


def build(force: bool = False, verbose: bool = True) -> Path:
    """Compile every HIP source for gfx950 and link libopenibl_amd.so (and libopenibl_amd_dbg.so, the
    same sources with the test hooks compiled in).  Returns the product's path.

    One process per GPU means several ranks may arrive here at once: the build is serialised with
    an exclusive file lock (the ranks that waited find the stamp current and return), and the
    library is linked to a temporary name and renamed into place, so no rank ever maps a
    half-written file."""
    if not force and is_current():
        return LIB_PATH
    BUILD_DIR.mkdir(parents=True, exist_ok=True)
    import fcntl
    with open(BUILD_DIR / "lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and is_current():
                return LIB_PATH
            return _build_locked(verbose, force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _parse_resource_usage(text: str) -> dict:
    """{mangled kernel name: {"VGPRs": n, "AGPRs": n, "ScratchSize": bytes per lane, "LDS": bytes,
    "Occupancy": waves per SIMD, "SGPRs": n}} from hipcc's kernel-resource-usage remarks."""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return out


def resource_usage() -> dict:
    """The report of the current build (builds first if the library is stale)."""
    build(verbose=False)
    p = BUILD_DIR / "resource_usage.json"
    if not p.exists():
        build(force=True, verbose=False)
    return json.loads(p.read_text())


def kernel_text() -> dict:
    """{kernel symbol: {"instructions": n, "mfma": matrix instructions in its text, "bytes": code size}} of the
    PRODUCT objects of the current build, from the gfx950 code objects themselves (llvm-objcopy the .hip_fatbin
    section out of each object, clang-offload-bundler --unbundle, llvm-objdump -d).  A hot kernel's text is
    part of what is measured: round 4 lost 4.5 % of the step to a kernel body the compiler had cloned around a
    diagnostic flag that was off (tests/test_abi.py pins the matrix-instruction counts).  Returns {} when the
    LLVM tools are not next to hipcc."""
    import re
    import tempfile
    build(verbose=False)
    cache = BUILD_DIR / "kernel_text.json"
    stamp = (BUILD_DIR / "stamp").read_text()
    if cache.exists():
        hit = json.loads(cache.read_text())
        if hit.get("stamp") == stamp:
            return hit["kernels"]
    bindir = None
    for cand in (Path(_hipcc()).resolve().parent, Path("/opt/rocm/lib/llvm/bin"), Path("/opt/rocm/llvm/bin")):
        if (cand / "llvm-objdump").exists() and (cand / "clang-offload-bundler").exists():
            bindir = cand
            break
    if bindir is None:
        return {}
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for src in sources():
            obj = BUILD_DIR / (src.stem + ".o")
            fat, co = Path(td) / (src.stem + ".fat"), Path(td) / (src.stem + ".co")
            r = subprocess.run([str(bindir / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)],
                               capture_output=True, text=True)
            if r.returncode != 0 or not fat.exists():
                continue
            r = subprocess.run([str(bindir / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                                f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}", f"--output={co}"],
                               capture_output=True, text=True)
            if r.returncode != 0:
                continue
            r = subprocess.run([str(bindir / "llvm-objdump"), "-d", "--no-show-raw-insn", str(co)],
                               capture_output=True, text=True)
            cur, first = None, 0
            for line in r.stdout.splitlines():
                m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
                if m:
                    cur = out.setdefault(m.group(2), {"instructions": 0, "mfma": 0, "bytes": 0})
                    first = int(m.group(1), 16)
                    continue
                m = re.match(r"^\s+(\S.*?)\s*// ([0-9A-Fa-f]+):", line)
                if m and cur is not None:
                    cur["instructions"] += 1
                    cur["mfma"] += 1 if "v_mfma" in m.group(1) else 0
                    cur["bytes"] = int(m.group(2), 16) - first
    cache.write_text(json.dumps({"stamp": stamp, "kernels": out}))
    return out


def _includes(src: Path, seen=None) -> list:
    """The in-tree headers a source reaches through #include "..." (csrc/ and include/), transitively."""
    import re
    seen = {} if seen is None else seen
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', src.read_text(), flags=re.M):
        for base in (src.parent, CSRC, INCLUDE):
            h = base / name
            if h.exists() and h not in seen:
                seen[h] = True
                _includes(h, seen)
                break
    return sorted(seen)


def _object_key(src: Path, dbg: bool) -> str:
    """What an object file depends on: its source, the headers it reaches, the flags."""
    h = hashlib.sha256()
    for p in [src, *_includes(src)]:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(CXXFLAGS + (["-DOIBL_DEBUG_HOOKS", *DBG_EXPERIMENT_FLAGS] if dbg else [])).encode())
    return h.hexdigest()


def _build_locked(verbose: bool, force: bool = False) -> Path:
    hipcc = _hipcc()
    srcs = sources()
    BUILD_DIR_DBG.mkdir(parents=True, exist_ok=True)
    jobs = [(s, BUILD_DIR / (s.stem + ".o"), False) for s in srcs] + \
           [(s, BUILD_DIR_DBG / (s.stem + ".o"), True) for s in srcs]

    def compile_one(job):
        src, obj, dbg = job
        # an object whose source, headers and flags are unchanged is kept (conv.hip alone is two minutes of hipcc);
        # its compiler remarks are kept next to it so that the resource report stays complete
        key, keyfile, remfile = _object_key(src, dbg), obj.with_suffix(".key"), obj.with_suffix(".remarks")
        if not force and obj.exists() and keyfile.exists() and remfile.exists() and keyfile.read_text() == key:
            return dbg, remfile.read_text()
        # -Rpass-analysis: the compiler's per-kernel register / scratch / LDS report, kept next to the
        # objects (build/resource_usage.json; tests/test_abi.py holds the hot kernels to their budgets)
        cmd = [hipcc, *CXXFLAGS, *(["-DOIBL_DEBUG_HOOKS", *DBG_EXPERIMENT_FLAGS] if dbg else []),
               "-Rpass-analysis=kernel-resource-usage", "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}{' (debug hooks)' if dbg else ''}:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[openibl_amd.build] compiled {src.name}{' (debug hooks)' if dbg else ''}", file=sys.stderr)
        remfile.write_text(r.stderr)
        keyfile.write_text(key)
        return dbg, r.stderr

    # the big translation units first, so that the pool is never left with one long job at the end
    jobs.sort(key=lambda j: -j[0].stat().st_size)
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(jobs))) as ex:
        results = list(ex.map(compile_one, jobs))
    remarks = "\n".join(text for dbg, text in results if not dbg)
    (BUILD_DIR / "resource_usage.json").write_text(json.dumps(_parse_resource_usage(remarks), indent=1))
    (BUILD_DIR_DBG / "resource_usage.json").write_text(json.dumps(
        _parse_resource_usage("\n".join(text for dbg, text in results if dbg)), indent=1))

    for lib, objdir in ((LIB_PATH, BUILD_DIR), (LIB_PATH_DBG, BUILD_DIR_DBG)):
        tmp = lib.with_name(lib.name + f".tmp{os.getpid()}")
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc",
               *[str(objdir / (s.stem + ".o")) for s in srcs], "-o", str(tmp)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            tmp.unlink(missing_ok=True)
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, lib)
        if verbose:
            print(f"[openibl_amd.build] linked {lib}", file=sys.stderr)
    (BUILD_DIR / "stamp").write_text(_digest())
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
