import csv,collections,sys
def load(f):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda:[0,0.0,0.0]); fam=collections.defaultdict(float)
    for r in rows:
        a=agg[(r['family'],r['label'])]; a[0]+=1; a[1]+=float(r['ms']); a[2]+=float(r['flops']); fam[r['family']]+=float(r['ms'])
    return agg,fam
a,fa=load(sys.argv[1]); b,fb=load(sys.argv[2])
print("families/fwd A:",{k:round(v/3,3) for k,v in fa.items()}, round(sum(fa.values())/3,3))
print("families/fwd B:",{k:round(v/3,3) for k,v in fb.items()}, round(sum(fb.values())/3,3))
d=[]
for k in set(a)|set(b):
    va=a.get(k,[0,0,0]); vb=b.get(k,[0,0,0]); d.append(((vb[1]-va[1])/3,k,va,vb))
d.sort()
for x in d[:12]+d[-6:]:
    k=x[1]; va=x[2]; vb=x[3]
    print(f"{k[0]} {k[1][:66]:66s} A us={1e3*va[1]/max(1,va[0]):7.1f} B us={1e3*vb[1]/max(1,vb[0]):7.1f} d/fwd={x[0]:+.3f}")
