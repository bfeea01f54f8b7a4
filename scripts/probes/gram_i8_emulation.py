"""CPU emulation behind cnmf_e_amd/csrc/gram_i8.hpp: the ring weights W from a covariance table of 24 / 28 / 32-bit fixed-point data, and from the base-256 digit-pair
sums with the low weight classes dropped, against the table of the fp32 data -- 40 sampled pixels of a 64 x 64 x 3000 synthetic video, ring radius 15.
Result (round 5): 32 bits 1.3e-8, classes p + r >= 2 1.3e-8, >= 3 4.7e-7, >= 4 7.9e-5."""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
from cnmf_e_amd import synth
d1=d2=64; T=3000; K=12; r=15
f=synth.make_factors(d1,d2,T,K,3)
Y=synth.make_video(f,np.float32)          # (d, T)?
print(Y.shape, Y.dtype)
Yd=Y.astype(np.float64)
if Yd.shape[0]==T: Yd=Yd.T
X=Yd-Yd.mean(1,keepdims=True)
X=X.astype(np.float32).astype(np.float64)   # the engine's centred fp32 video
def quant(X,bits):
    s=np.abs(X).max(1,keepdims=True)/(2.0**(bits-1)-1)
    return np.round(X/s)*s
# ring offsets
offs=[(dr,dc) for dc in range(-r-1,r+2) for dr in range(-r-1,r+2) if r<=np.hypot(dr,dc)<r+1]
print(len(offs))
rng=np.random.default_rng(0)
pix=[(int(a),int(b)) for a,b in zip(rng.integers(r+1,d1-r-1,40),rng.integers(r+1,d2-r-1,40))]
for bits in (24,28,32):
    Xq=quant(X,bits)
    errs=[]
    for (pr,pc) in pix:
        idx=[(pc+dc)*d1+(pr+dr) for dr,dc in offs]
        c=pc*d1+pr
        def solve(Z):
            R=Z[idx]; G=R@R.T; g=R@Z[c]; u=R.sum(1)
            n=len(idx); M=np.zeros((n+1,n+1)); M[:n,:n]=G; M[:n,n]=u; M[n,:n]=u; M[n,n]=T
            lam=1e-5*np.trace(M); rhs=np.concatenate([g,[Z[c].sum()]])
            return np.linalg.solve(M+lam*np.eye(n+1),rhs)[:n]
        w=solve(X); wq=solve(Xq)
        errs.append(np.abs(wq-w).max()/np.abs(w).max())
    print(bits, 'max rel err W', max(errs), 'median', np.median(errs))

# digits: balanced base-256, classes p+q >= 3 only
def digits(X):
    s=np.abs(X).max(1,keepdims=True)/(2.0**31-2.0**24)
    q=np.round(X/s).astype(np.int64)
    D=[]
    for p in range(4):
        d=((q+128)&255)-128
        q=(q-d)>>8
        D.append(d.astype(np.float64))
    assert np.all(q==0)
    return D,s
D,s=digits(X)
def gram_digits(idx_rows, idx_cols, minclass):
    G=0
    for p in range(4):
        for qq in range(4):
            if p+qq>=minclass:
                G=G+(256.0**(p+qq))*(D[p][idx_rows]@D[qq][idx_cols].T)
    return G*(s[idx_rows]*s[idx_cols].T)
for minclass in (0,2,3,4):
    errs=[]
    for (pr,pc) in pix:
        idx=[(pc+dc)*d1+(pr+dr) for dr,dc in offs]; c=pc*d1+pr
        R=X[idx]; G=R@R.T; g=R@X[c]; u=R.sum(1); n=len(idx)
        def slv(G,g):
            M=np.zeros((n+1,n+1)); M[:n,:n]=G; M[:n,n]=u; M[n,:n]=u; M[n,n]=T
            lam=1e-5*np.trace(M); rhs=np.concatenate([g,[X[c].sum()]])
            return np.linalg.solve(M+lam*np.eye(n+1),rhs)[:n]
        w=slv(G,g)
        Gq=gram_digits(idx,idx,minclass); gq=gram_digits(idx,[c],minclass)[:,0]
        wq=slv(Gq,gq)
        errs.append(np.abs(wq-w).max()/np.abs(w).max())
    print('classes >=',minclass,'max rel err W',max(errs),'median',np.median(errs))
