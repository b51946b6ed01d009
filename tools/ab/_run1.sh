python - <<EOF
import sys, numpy as np
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib
from oracle import Oracle
L=_lib.lib(); orc=Oracle()
rng=np.random.default_rng(5)
bad=0
for (w,h,n) in ((400,300,37),(533,400,9),(640,481,8),(300,200,13),(900,600,5),(321,240,10),(960,540,6),(250,250,7),(65,700,5),(799,64,6),(77,200,9),(130,131,9),(955,100,5)):
    imgs=rng.integers(0,256,(n,h,w),dtype=np.uint8)
    imgs=(imgs//4+np.linspace(0,180,w,dtype=np.float32)[None,None,:]).astype(np.uint8)
    L.cbh_set_tuning(b"hash_band_area",1); a=cbird_amd.dct_hash64_batch(imgs)
    o=orc.dcthash64_batch(imgs)
    print(w,h,n,"new==oracle",int((a==o).sum()),flush=True)
    bad+=int((a!=o).sum())
print("BAD",bad)
EOF
python tools/hash_sizes.py bytes=4e9 geos=400x300,533x400,300x200,800x600,900x600,320x240,960x540,720x540,600x400,480x360,200x150 ab=hash_band_area:0:1
