import sys, numpy as np, time
sys.path.insert(0, "/root/repo")
from crabml_amd import synth
from crabml_amd.engine import plan_clade_work, compress_patterns
model, nt, ns, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
case = synth.make_case(model, nt, ns, seed=3 if model=="ModelA" else 2)
s2p, w, first = compress_patterns(case.codes)
codes = np.ascontiguousarray(case.codes[:, first]).astype(np.int32)
n = codes.shape[1]
t=time.time()
whole = plan_clade_work(case.parent, case.leaf_node_ids, codes)
print(f"{model} {nt}x{ns}: patterns {n}; whole: clades {whole[0]} subpatterns {whole[1]} products {whole[2]} ({whole[2]/whole[3]:.3f} of whole walks) [{time.time()-t:.1f}s]")
def shares(order, label):
    c = codes[:, order]
    tiles = -(-n//64); per = -(-tiles//W)*64
    tot=0; mx=0
    for r in range(W):
        sl = c[:, r*per:min(n,(r+1)*per)]
        if sl.shape[1]==0: continue
        k = plan_clade_work(case.parent, case.leaf_node_ids, np.ascontiguousarray(sl))
        tot+=k[2]; mx=max(mx,k[2])
    print(f"  {label:34s}: sum of shares {tot/whole[2]:.3f} x whole, largest share {mx/whole[2]:.3f} x whole -> product speed-up {whole[2]/mx:.2f} of {W}")
shares(np.arange(n), "first appearance (today)")
# lexicographic by leaf rows in tree order
key = np.lexsort(codes[::-1])
shares(key, "lexicographic, leaf rows in order")
# consensus codon then lexicographic
cons = np.array([np.bincount(codes[:,p][codes[:,p]<61], minlength=61).argmax() for p in range(n)])
nd = np.array([(codes[:,p]!=cons[p]).sum() for p in range(n)])
shares(np.lexsort((nd, cons)), "consensus codon, then #differences")
shares(np.lexsort(tuple(codes[::-1]) + (cons,)), "consensus codon, then lexicographic")
# tree-aware: maximal subtrees of <= 64 leaves; which of them deviate from the consensus codon
par = np.asarray(case.parent); nn=len(par)
children=[[] for _ in range(nn)]
for v,p in enumerate(par):
    if p>=0: children[p].append(v)
leafset=set(int(x) for x in case.leaf_node_ids)
nleaf=np.zeros(nn,int)
order_nodes=[]
root=int(np.flatnonzero(par<0)[0])
st=[root]
while st:
    v=st.pop(); order_nodes.append(v); st.extend(children[v])
for v in reversed(order_nodes):
    nleaf[v]=1 if v in leafset else sum(nleaf[c] for c in children[v])
blocks=[]
st=[root]
while st:
    v=st.pop()
    if nleaf[v]<=64 or v in leafset: blocks.append(v)
    else: st.extend(children[v])
def leaves_below(v):
    out=[];s=[v]
    while s:
        x=s.pop()
        if x in leafset: out.append(x)
        else: s.extend(children[x])
    return out
rowof={int(l):i for i,l in enumerate(case.leaf_node_ids)}
blocks.sort(key=lambda v:-nleaf[v])
dev=np.zeros((len(blocks),n),dtype=np.int64)
for bi,v in enumerate(blocks):
    rows=[rowof[l] for l in leaves_below(v)]
    dev[bi]=((codes[rows]!=cons[None,:]) & (codes[rows]<61)).sum(axis=0)
mask_keys=tuple((dev[bi]>0).astype(np.int8) for bi in range(len(blocks)))[::-1]
shares(np.lexsort(mask_keys + (nd>0, cons)), "consensus, then deviating blocks")
shares(np.lexsort(mask_keys + (cons,)), "consensus, deviating-block mask only")
print(len(blocks),"blocks")
