"""video batch timing split (development aid)"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from cbird_amd import _lib, synth_video
from cbird_amd.video import DctVideoIndex, VideoIndex, VideoSearchParams
L = _lib.lib()
clips = synth_video.make_clips(10000, 300, seed=1234, subclip_frac=0.01, max_gap=8)
class M: pass
media = []
for i, (f, h) in enumerate(clips):
    m = M(); m.id, m.path, m.videoIndex = i + 1, f"c{i}", VideoIndex(f.tolist(), [int(x) for x in h]); media.append(m)
v = DctVideoIndex(); v.add(media)
p = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=30, minFramesNear=60)
v.findVideo(media[0], p)
needles = media[-2000:]
t0 = time.time(); r = v.find_videos_batch(needles, p); t1 = time.time()
print("batch total", t1 - t0)
# the ctypes call alone
f = np.concatenate([np.asarray(m.videoIndex.frames, np.int32) for m in needles])
h = np.concatenate([np.asarray(m.videoIndex.hashes, np.uint64) for m in needles])
o = np.zeros(len(needles) + 1, np.uint64); np.cumsum([len(m.videoIndex.frames) for m in needles], out=o[1:])
i = np.ascontiguousarray([m.id for m in needles], np.uint32)
from cbird_amd._lib import cbh_vmatch
buf = (cbh_vmatch * 16000)(); oo = np.zeros(len(needles) + 1, np.uint64)
st = _lib.cbh_stats()
t0 = time.time()
rc = L.cbh_vidx_find_videos_batch(v._h, f.ctypes.data, h.ctypes.data, o.ctypes.data, i.ctypes.data, len(needles), 5, 0, 30, 60, 1, buf, 16000, oo.ctypes.data)
t1 = time.time()
print("C call", t1 - t0, "rc", rc, "results", int(oo[-1]))
