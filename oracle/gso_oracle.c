/* filled in below */
#include "oracle.h"
