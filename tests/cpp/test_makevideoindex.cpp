// TEST SCAFFOLD: cbird_gpu::gpuMakeVideoIndex (the drop-in for Media::makeVideoIndex, src/media.cpp:925-1037) compiled
// against the mock headers and run on the device.  usage: test_makevideoindex frames.raw w h n threshold chunk stop_at
//   prints the index of the whole clip ("full"), then -- stop_at > 0 -- indexes the first stop_at frames, saves that
//   as a .vdx next to the raw file, loads it and resumes ("resumed"); tests/test_cpp_adapter.py compares both with the
//   oracle.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "videocontext.h"
#include "gpu_cvutil.h"

static void show(const char* tag, const VideoIndex& ix) {
  printf("%s %zu", tag, ix.frames.size());
  for (size_t i = 0; i < ix.frames.size(); ++i) printf(" %d:%" PRIu64, ix.frames[i], uint64_t(ix.hashes[i]));
  printf("\n");
}

int main(int argc, char** argv) {
  if (argc < 8) return 2;
  const char* raw = argv[1];
  const int w = atoi(argv[2]), h = atoi(argv[3]), n = atoi(argv[4]), thr = atoi(argv[5]), chunk = atoi(argv[6]);
  const int stopAt = atoi(argv[7]);
  int lastPercent = -1, calls = 0;
  {
    VideoContext video(raw, w, h, n, true);
    VideoIndex index;
    cbird_gpu::gpuMakeVideoIndex(video, thr, index, [&](int p) { lastPercent = p, ++calls; }, chunk);
    show("full", index);
    printf("progress %d %d\n", lastPercent, calls > 0);
  }
  if (stopAt > 0) {
    VideoIndex part;
    {
      VideoContext video(raw, w, h, stopAt, true);
      cbird_gpu::gpuMakeVideoIndex(video, thr, part, nullptr, chunk);
    }
    show("part", part);
    const std::string vdx = std::string(raw) + ".resume.vdx";
    part.save(vdx.c_str());
    VideoIndex loaded;
    loaded.load(vdx.c_str());
    {
      VideoContext video(raw, w, h, n, true);
      cbird_gpu::gpuMakeVideoIndex(video, thr, loaded, nullptr, chunk);
    }
    show("resumed", loaded);
    // a decoder that cannot seek: the old index is dropped and the video indexed from frame 0 (:934-937)
    VideoIndex again = part;
    {
      VideoContext video(raw, w, h, n, false);
      cbird_gpu::gpuMakeVideoIndex(video, thr, again, nullptr, chunk);
    }
    show("noseek", again);
  }
  return 0;
}
