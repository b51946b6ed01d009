#!/bin/bash
# Pin the OpenCV-dependent half of the hot path against the library cbird pins (OpenCV 2.4.13.7: cbird.pri:148-152, build
# recipe docker/build-opencv.sh).  Nothing in the build container can run this (no OpenCV, no network); it is written so
# that whoever can, cannot get it wrong:
#
#   tools/pin_with_opencv.sh <opencv-2.4.13.7-install-prefix> [<opencv-2.4.13.7-source-tree>]
#
#   1. builds tools/gen_golden_opencv.cpp against <prefix> (headers + core / imgproc / features2d) and checks that the
#      library says 2.4.13.x;
#   2. runs it and converts its text output into tests/golden/opencv_hash.npz (tools/opencv_golden_to_npz.py);
#   3. with the source tree: extracts rBRIEF's learned test pairs `bit_pattern_31_` (modules/features2d/src/orb.cpp) into
#      tests/golden/orb_bit_pattern_31.txt -- 1024 integers, what cbh_orb_set_pattern / cbird_amd.orb.load_pattern take;
#      without it the ORB descriptor goldens (record O) cannot be compared and stay skipped;
#   4. runs the tests that consume the goldens, restated stage by restated stage, and prints which stage is the FIRST to
#      disagree with OpenCV: blur / INTER_AREA (the 32x32 tile), cv::dct, cv::sum + threshold (the hash), BGR2GRAY,
#      INTER_LANCZOS4, the in-place keypoint squares, then pyramid resize / Gaussian / FAST / ORB keypoints + descriptors /
#      fastAtan2 / ellipse mask / BGR2Luv / kmeans.  The files to correct are named beside each stage.
#
# Exit code 0 = every stage that has goldens agrees (the "parity unpinned" notes in oracle/*.c, DESIGN.md and
# tests/test_opencv_golden.py can go); 1 = a stage disagrees (printed); 2 = could not build / run.
set -u
cd "$(dirname "$0")/.."
PREFIX="${1:-}"
SRC="${2:-}"
if [ -z "$PREFIX" ] || [ ! -d "$PREFIX/include/opencv2" ]; then
  echo "usage: $0 <opencv-2.4.13.7-install-prefix> [<opencv-source-tree>]   (no $PREFIX/include/opencv2)"; exit 2
fi
OUT=$(mktemp -d)
LIBS=""
for l in opencv_core opencv_imgproc opencv_features2d opencv_flann; do
  if ls "$PREFIX"/lib*/lib$l.* >/dev/null 2>&1; then LIBS="$LIBS -l$l"; fi
done
LIBDIR=$(dirname "$(ls "$PREFIX"/lib*/libopencv_core.* | head -1)")
echo "== 1. build against $PREFIX ($LIBS)"
g++ -O2 -std=c++11 tools/gen_golden_opencv.cpp -o "$OUT/gen_golden_opencv" -I"$PREFIX/include" -L"$LIBDIR" $LIBS \
    -Wl,-rpath,"$LIBDIR" || { echo "build failed"; exit 2; }
echo "== 2. generate"
"$OUT/gen_golden_opencv" > "$OUT/opencv_hash.txt" || { echo "generator failed"; exit 2; }
VER=$(grep -m1 '^V ' "$OUT/opencv_hash.txt" | cut -d' ' -f2)
echo "   library version: $VER"
case "$VER" in 2.4.13*) ;; *) echo "   NOT the version cbird pins (2.4.13.7): goldens from it prove nothing -- stopping"; exit 2;; esac
python3 tools/opencv_golden_to_npz.py "$OUT/opencv_hash.txt" tests/golden/opencv_hash.npz || exit 2
echo "   wrote tests/golden/opencv_hash.npz"
if [ -n "$SRC" ] && [ -f "$SRC/modules/features2d/src/orb.cpp" ]; then
  echo "== 3. bit_pattern_31_ from $SRC/modules/features2d/src/orb.cpp"
  python3 - "$SRC/modules/features2d/src/orb.cpp" <<'PY' || exit 2
import sys
sys.path.insert(0, ".")
from cbird_amd.orb import load_pattern
p = load_pattern(sys.argv[1])
open("tests/golden/orb_bit_pattern_31.txt", "w").write(" ".join(str(int(v)) for v in p) + "\n")
print("   wrote tests/golden/orb_bit_pattern_31.txt (1024 integers)")
PY
else
  echo "== 3. no source tree given: bit_pattern_31_ not extracted; ORB descriptor goldens stay unchecked"
fi
echo "== 4. compare, stage by stage"
python3 tools/pin_report.py
