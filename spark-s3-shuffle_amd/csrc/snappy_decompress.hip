// placeholder until the snappy decode kernel lands
#include "s3s_internal.h"
