"""Fingerprint of ONE kernel's machine code inside libmi_denoise.so -- pure Python, no GPU, no torch.

Why: bench.py's `roofline.traffic` / `valu_util` are hardware-counter figures, and PMC counters cannot be read inside an
un-profiled run; they are read back from profiles/r*_traffic.json / r*_utilisation.json (rocprofv3 --pmc passes of the same
command).  A kernel edited after those passes would silently carry stale counters into a bench line.  So the profile
summaries store the sha256 of the profiled kernel's code (tools/run_profiles.sh writes it on the GPU box, from the library
that was profiled), and bench.py recomputes it from the library it has LOADED: a mismatch turns the read-back figures into
null + the reason.

What is hashed: the kernel function's bytes (its .text range in the gfx950 code object, st_value .. st_value + st_size) followed
by its 64-byte kernel descriptor (`<name>.kd`: VGPR/SGPR/LDS allocation, which changes occupancy without changing a single
instruction).  Layout walked: host ELF section .hip_fatbin -> clang offload bundles (`__CLANG_OFFLOAD_BUNDLE__`, uncompressed as
hipcc 7.x emits them) -> the `hipv4-amdgcn-amd-amdhsa--gfx950` entry (an ELF64 code object) -> .symtab / .dynsym.
"""
import hashlib
import json
import struct
import sys

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# nlm_strip_kernel<SLO=-10, SHI=11, PLO=-3, PHI=4, R=8, NW=4, FMT=0 (RGBA32F), FUSED, !MULTI, !HALF>: bench.py's timed launch
# bilateral_kernel<R=8, P=2, NW=8, FMT=0 (RGBA32F), LINEAR, MODE=0, BilOne>: `bench.py --workload bilateral`'s launch
BENCH_KERNELS = {
    "nlm": "_ZN3mid16nlm_strip_kernelILin10ELi11ELin3ELi4ELi8ELi4ELi0ELb1ELb0ELb0ELi0EEEvNS_7NlmArgsE",
    "bilateral": "_ZN3mid16bilateral_kernelILi8ELi2ELi8ELi0ELb1ELi0ENS_6BilOneEEEvNS_7BilArgsET5_",
}


def _sections(elf):
    if elf[:4] != b"\x7fELF" or elf[4] != 2 or elf[5] != 1:
        raise ValueError("not a little-endian ELF64 image")
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    sec = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        sec.append({"name_off": name, "type": typ, "addr": addr, "off": off, "size": size, "link": link, "entsize": entsize})
    strtab = sec[shstrndx]
    for s in sec:
        end = elf.index(b"\0", strtab["off"] + s["name_off"])
        s["name"] = elf[strtab["off"] + s["name_off"]:end].decode()
    return sec


def gfx950_code_objects(lib_bytes):
    """The gfx950 ELF code objects bundled in the host library, one per translation unit."""
    fat = [s for s in _sections(lib_bytes) if s["name"] == ".hip_fatbin"]
    if not fat:
        raise ValueError("no .hip_fatbin section")
    lo, hi = fat[0]["off"], fat[0]["off"] + fat[0]["size"]
    out, p = [], lib_bytes.find(_MAGIC, lo, hi)
    while p >= 0:
        n, = struct.unpack_from("<Q", lib_bytes, p + len(_MAGIC))
        q = p + len(_MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", lib_bytes, q)
            q += 24
            triple = lib_bytes[q:q + tlen].decode()
            q += tlen
            if "gfx950" in triple and size:
                out.append(lib_bytes[p + off:p + off + size])
        p = lib_bytes.find(_MAGIC, p + len(_MAGIC), hi)
    return out


def _symbols(elf):
    sec = _sections(elf)
    syms = {}
    for s in sec:
        if s["type"] not in (2, 11):                       # SHT_SYMTAB, SHT_DYNSYM
            continue
        strs = sec[s["link"]]
        for i in range(s["size"] // 24):
            name, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", elf, s["off"] + 24 * i)
            if not name or shndx == 0 or shndx >= len(sec):
                continue
            end = elf.index(b"\0", strs["off"] + name)
            syms[elf[strs["off"] + name:end].decode()] = (sec[shndx], value, size)
    return syms


def kernel_sha256(lib_path, mangled):
    """sha256 hex of (function bytes + kernel descriptor) of `mangled` in the library's gfx950 code, and the function's size."""
    data = open(lib_path, "rb").read()
    for co in gfx950_code_objects(data):
        syms = _symbols(co)
        if mangled not in syms:
            continue
        h = hashlib.sha256()
        sizes = []
        for name in (mangled, mangled + ".kd"):
            if name not in syms:
                raise ValueError(f"{name} is missing from the code object that holds the kernel")
            s, value, size = syms[name]
            start = s["off"] + (value - s["addr"])
            h.update(co[start:start + size])
            sizes.append(size)
        return h.hexdigest(), sizes[0]
    raise ValueError(f"{mangled} is in no gfx950 code object of {lib_path}")


def fingerprint(lib_path, workload="nlm"):
    digest, size = kernel_sha256(lib_path, BENCH_KERNELS[workload])
    return {"kernel_symbol": BENCH_KERNELS[workload], "kernel_code_sha256": digest, "kernel_code_bytes": size,
            "what": "sha256 over the kernel function's bytes + its 64-byte kernel descriptor in libmi_denoise.so's gfx950 code object"}


if __name__ == "__main__":
    import os
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmi_denoise.so")
    out = fingerprint(path, "nlm")                              # top level: the headline kernel (as since round 5)
    out["workloads"] = {w: fingerprint(path, w) for w in BENCH_KERNELS}
    print(json.dumps(out, indent=1))
