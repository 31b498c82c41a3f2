/* slots.hip — slot layer (filled in below) */
#include "k_common.h"
extern "C" int init_acceleration_functions_mi355x(void* accel) { (void)accel; return M355_ERR_NO_DEVICE; }
