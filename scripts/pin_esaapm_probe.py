#!/usr/bin/env python3
"""Pins the oracle's restatement of esaapm / esahamming (oracle/vsapprox.c:
hamminghits, edithits -- patterns that are not cut, pieces with a threshold of
their own) against the REAL reference: random repetitive multi-sequence texts
with wildcards, reads of 6..33 symbols with up to K + 1 edit operations,
K = 1..3, edit and Hamming distance; `vmatch_ref -complete -e|-h K` and the
oracle must print the same list, order included.  Needs oracle/_ref.
usage: pin_esaapm_probe.py [SEED] [ROUNDS]"""
import os, shutil, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as H
H.build_oracle()
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
rounds=int(sys.argv[2]) if len(sys.argv)>2 else 10
ok=0
for rnd in range(rounds):
    nseq=int(rng.integers(1,4)); seqs=[]
    for s in range(nseq):
        n=int(rng.integers(3000,12000))
        t=rng.integers(0,4,n).astype(np.uint8)
        unit=rng.integers(0,4,int(rng.integers(20,80))).astype(np.uint8)
        for r in range(int(rng.integers(0,30))):
            p=int(rng.integers(0,n-len(unit))); u=unit.copy()
            for e in range(int(rng.integers(0,3))): u[int(rng.integers(0,len(u)))]=rng.integers(0,4)
            t[p:p+len(u)]=u
        for r in range(int(rng.integers(0,3))):
            ln=int(rng.integers(10,60)); a=int(rng.integers(0,n-ln)); t[a:a+ln]=np.resize(rng.integers(0,4,int(rng.integers(1,4))),ln)
        if rng.random()<0.6: t[rng.random(n)<0.002]=H.WILDCARD
        seqs.append(t)
    wd=tempfile.mkdtemp(prefix="pin_")
    H.write_fasta(wd+"/db.fna",[("s%d"%i,s) for i,s in enumerate(seqs)])
    H.run_mkvtree_ref(["-db","db.fna","-dna","-pl","-allout"],wd)
    idx=H.load_mkvtree_index(wd+"/db.fna")
    tis=idx.tis
    doedist=rng.random()<0.6
    m0=int(rng.integers(6,33))
    k=int(rng.integers(1,max(2,min(4,m0//3))))
    reads=[]
    for i in range(int(rng.integers(20,80))):
        m=m0 if rng.random()<0.7 else int(rng.integers(max(k+2,6),34))
        p=int(rng.integers(0,len(tis)-m)); q=tis[p:p+m].copy()
        q[q==H.SEPARATOR]=rng.integers(0,4)
        if rng.random()<0.8: q[q==H.WILDCARD]=rng.integers(0,4)
        for e in range(int(rng.integers(0,k+2))):
            kind,x=int(rng.integers(0,3)),int(rng.integers(0,len(q)))
            if kind==0 or not doedist: q[x]=(q[x]+1+rng.integers(0,3))%4 if q[x]<4 else 0
            elif kind==1 and len(q)>k+3: q=np.delete(q,x)
            else: q=np.insert(q,x,rng.integers(0,4))
        reads.append(q.astype(np.uint8))
    reads=[r for r in reads if len(r)>k and len(r)>=idx.prefixlength]
    H.write_fasta(wd+"/q.fna",[("q%d"%i,r) for i,r in enumerate(reads)])
    hq=H.Queries.from_list(reads)
    rc,lines,err=H.run_vmatch_ref(["-complete","-e" if doedist else "-h",str(k),"-q","q.fna","db.fna"],wd)
    want=H.parse_vmatch_lines(lines,approx=True)
    try:
        got=H.matches_as_ref(idx,H.oracle_approx(idx,hq,doedist,k))
    except H.OracleError as e:
        print("round",rnd,"oracle error",e,"ref rc",rc,err[:100]); continue
    # the reference's longest match reads past the end of the mapped text
    # (zero bytes of the page): matches starting within m + k of the end of
    # the whole text can come out one symbol longer there -- left out
    ends=idx.n-40
    def absolute(mm):
        ssp=np.concatenate([[0],idx.ssp.astype(np.int64)+1])
        return ssp[mm["dbseq"].astype(np.int64)]+mm["dbrel"].astype(np.int64)
    got=got[absolute(got)<ends]; want=want[absolute(want)<ends]
    same=np.array_equal(got,want)
    splits=sorted({(int(H.oracle_lib().orc_getoptsplit(int(doedist),10,4,idx.n,len(r),k))) for r in reads})
    print("round %d %s k=%d m0=%d n=%d reads=%d matches=%d splitsizes=%s : %s"%(rnd,"edist" if doedist else "hamming",k,m0,idx.n,len(reads),len(want),splits,"OK" if same else "MISMATCH"),flush=True)
    if not same:
        print(len(got),len(want)); 
        for a,b in zip(got,want):
            if a!=b: print("first diff",a,b); break
        qi=int(b[3]); r=reads[qi]; print("query",qi,"len",len(r),r.tolist())
        sys.exit(1)
    ok+=1
    shutil.rmtree(wd, ignore_errors=True)
print("all",ok,"ok")
