#include "nm_common.h"
