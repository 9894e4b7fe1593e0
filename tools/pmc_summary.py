"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_*) of bench.py into a text file."""
import csv, collections, sys
fetch, write, sq, steps, title = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
def agg(path):
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in rows:
        k = r['Kernel_Name'].split('(')[0].replace('void svsdf::', '').replace('svsdf::', '')
        a[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (k, r['Dispatch_Id']) not in seen:
            seen.add((k, r['Dispatch_Id'])); n[k] += 1
    return a, n
f, nf = agg(fetch); w, nw = agg(write); s, ns = agg(sq)
print(title)
print("separate passes: --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_* GRBM_GUI_ACTIVE; %d steps each (incl. warm-up)" % steps)
print("FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section): corrected column = 2x")
print("\nkernel                launches/step  FETCH MB/step  FETCHx2 MB/step  WRITE MB/step")
for k in sorted(f, key=lambda k: -f[k]['FETCH_SIZE']):
    if 'rocclr' in k: continue
    fs = f[k]['FETCH_SIZE']; ws = w.get(k, {}).get('WRITE_SIZE', 0)
    print(f"{k:20s} {nf[k]/steps:12.1f} {fs/steps/1024:14.2f} {2*fs/steps/1024:16.2f} {ws/steps/1024:14.2f}")
tf = sum(v['FETCH_SIZE'] for k, v in f.items() if 'k_solve' in k) / steps / 1024
tw = sum(v['WRITE_SIZE'] for k, v in w.items() if 'k_solve' in k) / steps / 1024
print(f"\nk_solve (all launches of one evaluation): HBM traffic = 2*FETCH + WRITE = {2*tf+tw:.1f} MB per evaluation")
print("\nkernel                waves/step  VALU/wave  SALU/wave  LDS/wave  SQ_BUSY_CYCLES/step  GRBM_GUI_ACTIVE/step")
for k, v in s.items():
    if 'rocclr' in k or not v.get('SQ_WAVES'): continue
    print(f"{k:20s} {v['SQ_WAVES']/steps:10.0f} {v['SQ_INSTS_VALU']/v['SQ_WAVES']:10.0f} {v['SQ_INSTS_SALU']/v['SQ_WAVES']:10.0f} {v['SQ_INSTS_LDS']/v['SQ_WAVES']:9.0f} {v['SQ_BUSY_CYCLES']/steps:20.4g} {v['GRBM_GUI_ACTIVE']/steps:20.4g}")
print(f"TRAFFIC_BYTES {int((2*tf+tw)*1e6)}")
# every kernel of the evaluation itself (k_*: the pipeline's own launches; not the one-time point upload -- k_points_* and the
# rocprim sort -- nor k_rbound at svsdf_create)
path = lambda k: k.startswith('k_') and not k.startswith('k_points') and not k.startswith('k_rbound')
af = sum(v['FETCH_SIZE'] for k, v in f.items() if path(k)) / steps / 1024
aw = sum(v['WRITE_SIZE'] for k, v in w.items() if path(k)) / steps / 1024
print(f"whole evaluation (every k_* launch of the path): HBM traffic = 2*FETCH + WRITE = {2*af+aw:.1f} MB per evaluation")
print(f"TRAFFIC_TOTAL_BYTES {int((2*af+aw)*1e6)}")
