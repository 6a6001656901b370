#!/bin/bash
# Register / LDS footprint of every kernel in libvbz_hip.so (from the code object's metadata; no GPU needed):
#   bash tools/kernel_resources.sh [name-filter]
set -u
lib=$(dirname "$0")/../vbz_compression_amd/lib/libvbz_hip.so
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$lib" --output="$tmp/co" --unbundle 2>/dev/null \
  || /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$tmp/fat" "$lib" 2>/dev/null
if [ ! -s "$tmp/co" ]; then
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fat" --output="$tmp/co" --unbundle
fi
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/co" | python3 -c '
import sys, re
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = []
for line in sys.stdin:
    m = re.match(r"\s*(-?)\s*\.(\w+):\s*(.*)", line)
    if not m: continue
    dash, k, v = m.group(1), m.group(2), m.group(3).strip()
    if dash and k == "agpr_count":      # a kernel entry begins (keys are in alphabetical order)
        cur = {}; rows.append(cur)
    if cur is not None and k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count") and k not in cur:
        cur[k] = v
import subprocess
for r in rows:
    if "vgpr_count" not in r: continue
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("vbzhip::(anonymous namespace)::", "").replace("void ", "")
    if flt and flt not in name: continue
    print("%-70s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %5s spill v%s s%s" % (name[:70], r.get("vgpr_count"), r.get("agpr_count", "0"), r.get("sgpr_count"), r.get("group_segment_fixed_size"), r.get("private_segment_fixed_size"), r.get("vgpr_spill_count", "0"), r.get("sgpr_spill_count", "0")))
' "${1:-}"
rm -rf "$tmp"
