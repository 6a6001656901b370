#!/bin/bash
# Register / LDS footprint of every kernel in libvbz_hip.so (from the code objects' metadata; no GPU needed):
#   bash tools/kernel_resources.sh [name-filter]
set -u
lib=$(dirname "$0")/../vbz_compression_amd/lib/libvbz_hip.so
tmp=$(mktemp -d)
objcopy --dump-section .hip_fatbin="$tmp/fat" "$lib" 2>/dev/null
python3 - "$tmp" "${1:-}" <<'PY'
import re, subprocess, sys
tmp, flt = sys.argv[1], sys.argv[2]
fat = open(tmp + "/fat", "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
starts = [m.start() for m in re.finditer(magic, fat)] + [len(fat)]
rows = []
for i in range(len(starts) - 1):
    part = tmp + "/b%d" % i
    open(part, "wb").write(fat[starts[i]:starts[i + 1]])
    co = part + ".co"
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part,
                    "--output=" + co, "--unbundle"], capture_output=True)
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    cur = None
    for line in notes.splitlines():
        m = re.match(r"\s*(-?)\s*\.(\w+):\s*(.*)", line)
        if not m:
            continue
        dash, k, v = m.group(1), m.group(2), m.group(3).strip()
        if dash and k == "agpr_count":      # a kernel entry begins (keys are in alphabetical order)
            cur = {}
            rows.append(cur)
        if cur is not None and k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                                     "vgpr_spill_count", "sgpr_spill_count") and k not in cur:
            cur[k] = v
for r in rows:
    if "vgpr_count" not in r:
        continue
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("vbzhip::(anonymous namespace)::", "").replace("void ", "")
    if flt and flt not in name:
        continue
    print("%-72s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %5s spill v%s s%s" % (name[:72], r.get("vgpr_count"), r.get("agpr_count", "0"), r.get("sgpr_count"),
          r.get("group_segment_fixed_size"), r.get("private_segment_fixed_size"), r.get("vgpr_spill_count", "0"), r.get("sgpr_spill_count", "0")))
PY
rm -rf "$tmp"
