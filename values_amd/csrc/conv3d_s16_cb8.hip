// The 8-channel-chunk instances of conv3d_s16.hip (x-pair layers included) as their own translation unit: see S16_PART there.
#define S16_PART 1
#include "conv3d_s16.hip"
