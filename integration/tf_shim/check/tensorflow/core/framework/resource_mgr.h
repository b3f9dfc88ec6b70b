// NOT TensorFlow: forwards to the declaration stub of `make check` (see tf_decl_stub.h).
#include "tf_decl_stub.h"
