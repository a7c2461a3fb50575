#include "wsa_oracle.h"
