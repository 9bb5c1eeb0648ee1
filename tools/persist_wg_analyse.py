import sys,re
import numpy as np
rows=[]
for l in open(sys.argv[1]):
    m=re.match(r"persist wg (\d+): operator ([\d.]+) us gather ([\d.]+) us entries (\d+) imports (\d+) exports (\d+)",l)
    if m: rows.append([float(x) for x in m.groups()])
a=np.array(rows)
# two solves printed? take last G
G=int(a[:,0].max())+1
a=a[-G:]
op,ga,en,im,ex=a[:,1],a[:,2],a[:,3],a[:,4],a[:,5]
print("G",G,"op mean %.2f max %.2f min %.2f"%(op.mean(),op.max(),op.min()),"gather mean %.2f"%ga.mean())
print("entries mean %.0f max %.0f min %.0f"%(en.mean(),en.max(),en.min()))
print("corr op~entries %.3f op~imports %.3f op~exports %.3f"%(np.corrcoef(op,en)[0,1],np.corrcoef(op,im)[0,1],np.corrcoef(op,ex)[0,1]))
A=np.stack([np.ones(G),en,im,ex],1)
co,res,_,_=np.linalg.lstsq(A,op,rcond=None)
print("fit op = %.2f + %.3g*entries + %.3g*imports + %.3g*exports; resid std %.2f"%(co[0],co[1],co[2],co[3],(op-A@co).std()))
for x in range(8):
    sel=(np.arange(G)%8)==x
    print("xcd",x,"op mean %.2f max %.2f entries %.0f"%(op[sel].mean(),op[sel].max(),en[sel].mean()))
idx=np.argsort(-op)[:10]
for i in idx: print("slow wg",int(a[i,0]),op[i],en[i],im[i],ex[i])
