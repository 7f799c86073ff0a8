// Provider selection, mirroring slimt/QMM.cc:3-34: exactly one provider is
// compiled in; here it is always the HIP one.
#include "QMM.hh"

#define SLIMT_HAS_HIP 1
#include "qmm/Hip.inl.cc"
