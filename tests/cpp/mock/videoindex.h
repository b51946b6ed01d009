// mock/videoindex.h -- TEST SCAFFOLD ONLY: VideoIndex lives in mock/index.h
#pragma once
#include "index.h"
