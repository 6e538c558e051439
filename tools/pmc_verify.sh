export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_h2c; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $out/p -- python tools/verify_breakdown.py > $out/log.txt 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/prof_h2c/p/**/*counter_collection.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
by=collections.OrderedDict()
for r in rows:
    d=int(r['Dispatch_Id']); by.setdefault(d,{'k':r['Kernel_Name'],'grid':r['Grid_Size'],'dur':int(r['End_Timestamp'])-int(r['Start_Timestamp'])})[r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(by)
# last verify call: print the last 45 dispatches
for d in ids[-70:]:
    x=by[d]; w=x.get('SQ_WAVES',1) or 1
    print(d,x['k'][:22],x['grid'],'%.3fms'%(x['dur']/1e6),'waves',int(w),'valu/w %.0f'%(x.get('SQ_INSTS_VALU',0)/w),'wavecyc/w %.0f'%(x.get('SQ_WAVE_CYCLES',0)/w),'wait/w %.0f'%(x.get('SQ_WAIT_ANY',0)/w),'vmrd/w %.0f'%(x.get('SQ_INSTS_VMEM_RD',0)/w),'vmwr/w %.0f'%(x.get('SQ_INSTS_VMEM_WR',0)/w))
PY
