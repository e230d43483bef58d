#include "petsc_types_stub.h"
