// mock/videocontext.h -- TEST SCAFFOLD ONLY: the part of cbird's VideoContext (src/videocontext.h:49-72,151-172)
// that Media::makeVideoIndex touches, reading raw grey frames from a file instead of a decoder.
#pragma once
#include <cstdio>
#include <vector>

#include "index.h"

class VideoContext {
 public:
  struct Metadata {
    float frameRate = 0.0f;
    int duration = 0;
  };
  VideoContext(const char* rawPath, int w, int h, int nFrames, bool seekable) : _w(w), _h(h), _n(nFrames), _seekable(seekable) {
    _f = fopen(rawPath, "rb");
    _metadata.frameRate = 25.0f;
    _metadata.duration = (nFrames + 24) / 25;
  }
  ~VideoContext() {
    if (_f) fclose(_f);
  }
  bool seek(int frame) {
    if (!_seekable || !_f || frame < 0 || frame > _n) return false;
    _pos = frame;
    return fseek(_f, long(frame) * _w * _h, SEEK_SET) == 0;
  }
  bool nextFrame(cv::Mat& outImg) {
    if (!_f || _pos >= _n) return false;
    if (outImg.rows != _h || outImg.cols != _w) outImg = cv::Mat(_h, _w);
    for (int y = 0; y < _h; ++y)
      if (fread(outImg.ptr<uint8_t>(y), 1, size_t(_w), _f) != size_t(_w)) return false;
    ++_pos;
    return true;
  }
  const Metadata& metadata() const { return _metadata; }
  int width() const { return _w; }
  int height() const { return _h; }

 private:
  FILE* _f = nullptr;
  int _w, _h, _n, _pos = 0;
  bool _seekable;
  Metadata _metadata;
};
