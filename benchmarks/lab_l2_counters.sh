cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pa -o a -- $R/benchmarks/bin/halo_lab 4 > /dev/null 2> /tmp/pa.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d /tmp/pb -o b -- $R/benchmarks/bin/halo_lab 4 > /dev/null 2> /tmp/pb.err
python3 - <<'PY' > $R/gpurun_out/r03_lab_l2.txt 2>&1
import csv,glob,collections
for d in ('/tmp/pa','/tmp/pb'):
    f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    print(d,f)
    if not f: continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name']
        if 'wgrad_halo' not in k and 'conv_halo16' not in k: continue
        key=(k[k.find('wgrad_halo') if 'wgrad_halo' in k else k.find('conv_halo16'):][:48], r['Grid_Size'])
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key,v in agg.items():
        print(key, {c:(len(x), sum(x)/len(x)) for c,x in v.items()})
PY
head -30 /tmp/pa.err >> $R/gpurun_out/r03_lab_l2.txt
