"""Print VGPR / spill / scratch / LDS / occupancy per kernel of one .hip file (hipcc remarks)."""
import re, subprocess, sys
src = sys.argv[1]
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: (?:\s*)(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
demangle = subprocess.run(["c++filt"], input="\n".join(x["name"] for x in rows), capture_output=True, text=True).stdout.splitlines()
for x, d in zip(rows, demangle):
    d = re.sub(r"\(anonymous namespace\)::", "", d); d = d.split("(")[0][:70]
    print(f"{d:70s} vgpr={x.get('VGPRs','?'):>4} agpr={x.get('AGPRs','?'):>3} spill={x.get('VGPR Spill','?'):>3} scratch={x.get('ScratchSize [bytes/lane]','?'):>5} lds={x.get('LDS Size [bytes/block]','?'):>6} occ={x.get('Occupancy [waves/SIMD]','?')}")
