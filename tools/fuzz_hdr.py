#!/usr/bin/env python3
"""Corrupted Radiance .hdr files (flat and run-length encoded) through load_hdr (run under the sanitizer build of
librt_host.so: tools/sanitize.sh).  usage: tools/fuzz_hdr.py <seed> <files>"""
import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import raytracing_amd as rta
import test_cpu_image_cli as T
import pathlib
rng=np.random.default_rng(int(sys.argv[1])); N=int(sys.argv[2])
tmp=pathlib.Path('/tmp/fz_hdr'); tmp.mkdir(exist_ok=True)
ok=bad=0
for it in range(N):
    w,h=int(rng.integers(8,40)),int(rng.integers(1,12))
    px=T._rgbe(rng,h,w)
    p=tmp/'a.hdr'
    T._write_hdr(p, px, bool(it&1))
    b=bytearray(p.read_bytes())
    m=it%3
    if m==0:
        for _ in range(int(rng.integers(1,6))): b[int(rng.integers(len(b)))]=int(rng.integers(256))
    elif m==1:
        b=b[:int(rng.integers(1,len(b)))]
    else:
        q=int(rng.integers(len(b))); b[q:q]=bytes(rng.integers(0,256,int(rng.integers(1,30)),dtype=np.uint8))
    p.write_bytes(bytes(b))
    sc=rta.Scene()
    if sc.load_env_hdr(str(p)): ok+=1
    else: bad+=1
sys.stderr.write(f"\nRESULT loaded {ok} refused {bad}\n")
