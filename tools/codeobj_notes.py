#!/usr/bin/env python3
"""tools/codeobj_notes.py [libacdsp.so] [filter] -- per-kernel resources straight from the code objects embedded in the library.

Pure Python (no llvm tools): finds every clang offload bundle in the file, takes its gfx950 code objects, reads the
NT_AMDGPU_METADATA note (msgpack) of each and prints / returns one record per kernel: VGPRs, AGPRs, SGPRs, scratch bytes per lane
(`.private_segment_fixed_size`), spilled VGPRs / SGPRs, static LDS.  `tests/test_abi.py::test_no_kernel_uses_scratch` runs
`kernels()` over the shipped library and fails on scratch outside its allow-list; the same numbers as
`llvm-readelf --notes` on the extracted bundles."""
import os
import struct
import subprocess
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _bundles(blob):
    """(triple, bytes) of every entry of every uncompressed offload bundle in the file."""
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        (n,) = struct.unpack_from("<Q", blob, pos + 24)
        q = pos + 32
        if n == 0 or n > 64:            # the magic string inside some unrelated data
            pos += 24
            continue
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode("ascii", "replace")
            q += 24 + tl
            yield triple, blob[pos + off:pos + off + size]
        pos += 24


def _notes(elf):
    """NT_AMDGPU_METADATA (type 32, owner AMDGPU) payloads of a 64-bit little-endian ELF."""
    if elf[:4] != b"\x7fELF":
        return
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:   # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        q, end = off, off + size
        while q + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, q)
            q += 12
            name = elf[q:q + namesz]
            q += (namesz + 3) & ~3
            desc = elf[q:q + descsz]
            q += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):
                yield desc


def kernels(path):
    """List of dicts: name (mangled), vgpr, agpr, sgpr, scratch, vgpr_spill, sgpr_spill, lds, max_wg."""
    blob = open(path, "rb").read()
    out = []
    for triple, obj in _bundles(blob):
        if "gfx950" not in triple:
            continue
        for desc in _notes(obj):
            md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in md.get("amdhsa.kernels", []):
                out.append({
                    "name": k.get(".name", "?"),
                    "vgpr": k.get(".vgpr_count", -1), "agpr": k.get(".agpr_count", 0), "sgpr": k.get(".sgpr_count", -1),
                    "scratch": k.get(".private_segment_fixed_size", 0),
                    "vgpr_spill": k.get(".vgpr_spill_count", 0), "sgpr_spill": k.get(".sgpr_spill_count", 0),
                    "lds": k.get(".group_segment_fixed_size", 0), "max_wg": k.get(".max_flat_workgroup_size", 0),
                })
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return [s.replace("void acdsp::", "").split("(")[0] for s in r.stdout.split("\n")[:len(names)]]
    except Exception:
        return list(names)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ac_dsp_amd", "lib", "libacdsp.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    ks = kernels(lib)
    names = demangle([k["name"] for k in ks])
    bad = 0
    for k, dn in sorted(zip(ks, names), key=lambda t: t[1]):
        if flt and flt not in dn:
            continue
        bad += k["scratch"] > 0
        print("%s %-64s vgpr %3d agpr %3d sgpr %3d scratch %4d spill %3d/%-3d lds %6d" % (
            "!" if k["scratch"] > 0 else " ", dn[:64], k["vgpr"], k["agpr"], k["sgpr"], k["scratch"], k["vgpr_spill"], k["sgpr_spill"], k["lds"]))
    print("# %d kernels, %d with scratch" % (len(ks), bad))
