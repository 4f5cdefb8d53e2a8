// The two-row-tile 16-channel-chunk instances of conv3d_s16.hip as their own translation unit: see S16_PART there.
#define S16_PART 2
#include "conv3d_s16.hip"
