#!/usr/bin/env python3
"""Corrupted copies of tests/golden/cornellbox.glb through Scene::load_from_gltf + build (run under the sanitizer build of
librt_host.so: tools/sanitize.sh).  usage: tools/fuzz_gltf.py <seed> <files>"""
import sys, os, struct, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import raytracing_amd as rta
src=open('/root/repo/tests/golden/cornellbox.glb','rb').read()
json_len=struct.unpack_from('<I',src,12)[0]
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
N=int(sys.argv[2]) if len(sys.argv)>2 else 200
ok=bad=0
for it in range(N):
    b=bytearray(src)
    mode=it%4
    if mode==0:   # corrupt JSON bytes (digits mostly)
        for _ in range(int(rng.integers(1,4))):
            p=20+int(rng.integers(json_len)); b[p]=int(rng.integers(48,58))
    elif mode==1: # corrupt header / chunk lengths
        p=int(rng.integers(0,20)); b[p]=int(rng.integers(256))
    elif mode==2: # corrupt binary chunk (indices / floats)
        for _ in range(int(rng.integers(1,20))):
            p=20+json_len+8+int(rng.integers(len(b)-20-json_len-8)); b[p]=int(rng.integers(256))
    else:         # truncate
        b=b[:int(rng.integers(12,len(b)))]
    path=f'/tmp/fuzz_{os.getpid()}.glb'
    open(path,'wb').write(bytes(b))
    sc=rta.Scene()
    try:
        sc.load_gltf(path)
        sc.build(); ok+=1
    except rta.RtError:
        bad+=1
print('loaded',ok,'refused',bad)
