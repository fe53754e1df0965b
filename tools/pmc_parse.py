import csv, collections, glob, sys
d=sys.argv[1]
cc=(glob.glob(d+'/*/*_counter_collection.csv')+glob.glob(d+'/*_counter_collection.csv'))[0]
rows=list(csv.DictReader(open(cc)))
acc=collections.defaultdict(lambda: collections.defaultdict(float))
names={}
for r in rows:
    if sys.argv[2] in r['Kernel_Name']:
        acc[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value'])
        names[r['Dispatch_Id']]=r['Kernel_Name'][18:52]+' vgpr '+r['VGPR_Count']
kt={r['Dispatch_Id']:(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(cc.replace('counter_collection','kernel_trace')))}
for d in sorted(acc, key=int)[-4:]:
    a=acc[d]; us=kt[d]
    out=[d, names[d], '%.0f us' % us]
    if 'GRBM_GUI_ACTIVE' in a:
        out.append('clk %.2f' % (a['GRBM_GUI_ACTIVE']/8/us/1e3))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in a: out.append('mfma_busy %.3f' % (a['SQ_VALU_MFMA_BUSY_CYCLES']/1024/(a['GRBM_GUI_ACTIVE']/8)))
    wc=a.get('SQ_WAVE_CYCLES',0)
    for k,v in a.items():
        if k.startswith('SQ_') and k not in ('SQ_WAVE_CYCLES','SQ_VALU_MFMA_BUSY_CYCLES') and wc: out.append('%s/wc %.3f' % (k[3:], v/wc))
        elif not k.startswith('SQ_') and k!='GRBM_GUI_ACTIVE': out.append('%s %.4g' % (k, v))
    print(' '.join(map(str,out)))
