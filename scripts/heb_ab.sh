for v in ${HEB_VARIANTS:-base hx1 hx2 hx3 old}; do L=$PWD/scripts/bin/libmcpc_$v.so; [ $v = base ] && L=$PWD/montecarlopredictivecoding_amd/libmcpc.so; [ $v = old ] && L=$PWD/scripts/bin/libmcpc_64db936.so
(cd /tmp && export TMPDIR=/tmp && FLUSH_TUNINGS="no_overlap=1,slot_cap=128" MCPC_LIB=$L rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4e/fa_$v -o stats -- python3 $GRAFT_REPO_ROOT/scripts/flush_alone.py 384 6000 > /dev/null 2>&1); g=$(find gpurun_out/r4e/fa_$v -name "*kernel_trace.csv" | head -1); python3 - "$g" $v <<EOF2
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "heb" in n: d[n.split("mcpc::")[1][:28]+" grid="+r.get("Grid_Size_X","?")].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print(sys.argv[2], " | ".join(f"{k}: {sorted(v)[len(v)//2]:7.1f}" for k,v in sorted(d.items())))
EOF2
rm -rf gpurun_out/r4e/fa_$v; done
