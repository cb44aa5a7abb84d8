#!/bin/bash
# tools/kres.sh <file.hip> [flags]: registers, spills, scratch and LDS of every kernel in the file (hipcc -Rpass-analysis=kernel-resource-usage)
F=$1; shift
cd /root/repo/utree_amd/csrc
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Rpass-analysis=kernel-resource-usage "$@" -c $F -o /tmp/kres.o 2>&1 | python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for ln in sys.stdin:
    m=re.search(r"remark: +(Function Name|[A-Za-z ]+[A-Za-z\]\[/ ]*): (.*?) \[-Rpass", ln)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2).strip()
    if k=="Function Name":
        cur={"name":v}; rows.append(cur)
    elif cur is not None: cur[k]=v
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    n=re.sub(r"\(.*","",n.replace("(anonymous namespace)::",""))
    print("%-60s VGPR %3s AGPR %3s SGPR %3s spill %s/%s scratch %s occ %s LDS %s" % (n.replace("void ","")[:60], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill","?"), r.get("SGPRs Spill","?"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
