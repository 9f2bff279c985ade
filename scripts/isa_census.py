"""Instruction census of one kernel in the saved ISA (`make -C ribotricer_amd/csrc asm` first).
usage: python scripts/isa_census.py [kernel-substring]"""
import collections
import os
import sys

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(here, "ribotricer_amd", "csrc", "_asm", "ribophase-hip-amdgcn-amd-amdhsa-gfx950.s")
want = sys.argv[1] if len(sys.argv) > 1 else "k_tile_score"
text = open(path).read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith("_ZN2rp") and want in l.split(":")[0])
end = next(i for i in range(start, len(text)) if text[i].strip().startswith("s_endpgm"))
ops = [l.split()[0] for l in (x.strip() for x in text[start + 1:end]) if l and l[0] not in ".;" and not l.endswith(":")]
count = collections.Counter(ops)
valu = sum(v for k, v in count.items() if k.startswith("v_"))
print(f"{text[start].split(':')[0]}: {len(ops)} instructions, {valu} VALU")
for k, v in count.most_common(45):
    print(f"  {k:28s}{v}")
for l in text[end:end + 400]:
    if any(t in l for t in ("next_free_vgpr", "group_segment_fixed_size", "private_segment_fixed_size")):
        print(l.strip())
    if ".end_amdhsa_kernel" in l:
        break
