#!/usr/bin/env python3
"""VALID gzip files of many shapes through the parallel inflate (host/pgzip.cpp) against zlib, via tests/host/test_pgzip (CPU only):
FASTQ / FASTA / random printable / extremely repetitive text, levels 1-9, the Z_FILTERED strategy, streams with sync and full
flushes (empty stored blocks, short fixed-Huffman blocks), two members; chunk sizes from 8 KB (chunks shorter than the 32 KB window)
to 256 KB, 2-8 threads.  Every file must come out byte for byte as zlib inflates it ("OK") or be declined before anything was
handed on ("NA").   python tools/fuzz_pgzip.py [seed=1] [n=300]"""
import gzip, zlib, subprocess, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, 'tests', 'host', 'test_pgzip')
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad=0; ok=0; na=0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    kind=int(rng.integers(0,5))
    n=int(rng.integers(200,6000))
    if kind==0: data=b"".join(b"@r%d\n"%i+bytes(rng.choice(np.frombuffer(b"ACGT",dtype=np.uint8),int(rng.integers(20,200))))+b"\n+\n"+b"I"*int(rng.integers(20,200))+b"\n" for i in range(n))
    elif kind==1: data=b"".join(b">s%d\n"%i+bytes(rng.choice(np.frombuffer(b"ACGTacgtN",dtype=np.uint8),int(rng.integers(20,400))))+b"\n" for i in range(n))
    elif kind==2: data=b"".join(b"@r\n"+bytes(rng.choice(np.frombuffer(b"ACGT",dtype=np.uint8),80))+b"\n+\n"+bytes(rng.integers(33,74,80,dtype=np.uint8))+b"\n" for i in range(n))
    elif kind==3: data=(b"ACGT"*50+b"\n")*n          # extremely repetitive: long matches, length 258 codes
    else: data=b"".join(bytes(rng.integers(32,127,int(rng.integers(1,300)),dtype=np.uint8))+b"\n" for i in range(n))
    level=int(rng.integers(1,10))
    mode=int(rng.integers(0,4))
    if mode==0: z=gzip.compress(data,level,mtime=0)
    elif mode==1:
        c=zlib.compressobj(level,zlib.DEFLATED,31,9,zlib.Z_FILTERED); z=c.compress(data)+c.flush()
    elif mode==2:
        c=zlib.compressobj(level,zlib.DEFLATED,31); z=b""
        pos=0
        while pos<len(data):
            step=int(rng.integers(1000,200000)); z+=c.compress(data[pos:pos+step]); pos+=step
            if rng.random()<0.5: z+=c.flush(zlib.Z_SYNC_FLUSH if rng.random()<0.7 else zlib.Z_FULL_FLUSH)
        z+=c.flush()
    else:
        k=len(data)//2; z=gzip.compress(data[:k],level,mtime=0)+gzip.compress(data[k:],level,mtime=0)
    open('/tmp/v.gz','wb').write(z)
    ch=str(int(rng.choice([8192,16384,32768,65536,262144])))
    th=str(int(rng.choice([2,3,4,8])))
    o=subprocess.run([exe,'/tmp/v.gz',th,ch],stdout=subprocess.PIPE,stderr=subprocess.PIPE).stdout.decode().strip()
    if o.startswith("OK"): ok+=1
    elif o.startswith("NA"): na+=1
    else:
        bad+=1; print("BAD", it, kind, level, mode, ch, th, len(data), len(z), o[:80]); os.system("cp /tmp/v.gz /tmp/bad_%d.gz"%it)
print("ok",ok,"na",na,"bad",bad)
sys.exit(1 if bad else 0)
