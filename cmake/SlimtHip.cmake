# WITH_HIP provider fragment for slimt's top-level CMakeLists.txt, in the idiom of its other
# providers (reference CMakeLists.txt:16-19 options, :125-144 provider blocks):
#
#   option(WITH_HIP "Use the MI355X (gfx950) HIP backend" OFF)
#   if(WITH_HIP)
#     include(<this repo>/cmake/SlimtHip.cmake)      # defines the target `slimt_hip`
#     list(APPEND SLIMT_PRIVATE_LIBS slimt_hip)
#     list(APPEND SLIMT_COMPILE_DEFINITIONS SLIMT_HAS_HIP)
#   endif(WITH_HIP)
#
# and in slimt/QMM.cc the provider selection gains `#elif defined(SLIMT_HAS_HIP)` with
# `#include "slimt/qmm/Hip.inl.cc"` (the file is
# slimt_amd/host/qmm/Hip.inl.cc).
#
# Inputs:  SLIMT_HIP_ROOT      this repository (default: the directory above this file)
#          SLIMT_HIP_PREBUILT  path of an already built libslimt_hip.so to import instead of
#                              compiling the kernels (e.g. slimt_amd/lib/libslimt_hip.so)
# Output:  target `slimt_hip` (shared library + include directory of slimt_hip.h)
#
# The kernels are compiled by hipcc for gfx950 only, with the float contract of
# slimt_amd/build.py: no contraction (fused ops are explicit fmaf), IEEE division / sqrt.
if(NOT DEFINED SLIMT_HIP_ROOT)
  get_filename_component(SLIMT_HIP_ROOT "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
endif()
set(SLIMT_HIP_CSRC "${SLIMT_HIP_ROOT}/slimt_amd/csrc")
set(SLIMT_HIP_INCLUDE "${SLIMT_HIP_ROOT}/include")

if(SLIMT_HIP_PREBUILT)
  add_library(slimt_hip SHARED IMPORTED GLOBAL)
  set_target_properties(slimt_hip PROPERTIES IMPORTED_LOCATION "${SLIMT_HIP_PREBUILT}"
                                             IMPORTED_NO_SONAME TRUE)
  target_include_directories(slimt_hip INTERFACE "${SLIMT_HIP_INCLUDE}")
else()
  find_program(SLIMT_HIPCC hipcc HINTS /opt/rocm/bin ENV ROCM_PATH PATH_SUFFIXES bin REQUIRED)
  set(SLIMT_HIP_SOURCES kernels.hip gemm_tile.hip decode_kernels.hip decode_fused.hip encode_fused.hip encode_wide.hip encode_tall.hip shortlist.hip engine.cpp)
  set(SLIMT_HIP_HEADERS kernels.h engine.h device_common.h shortlist_device.h)
  list(TRANSFORM SLIMT_HIP_SOURCES PREPEND "${SLIMT_HIP_CSRC}/")
  list(TRANSFORM SLIMT_HIP_HEADERS PREPEND "${SLIMT_HIP_CSRC}/")
  set(SLIMT_HIP_SO "${CMAKE_CURRENT_BINARY_DIR}/libslimt_hip.so")
  add_custom_command(
    OUTPUT "${SLIMT_HIP_SO}"
    COMMAND "${SLIMT_HIPCC}" -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off
            -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero
            -x hip ${SLIMT_HIP_SOURCES} -I "${SLIMT_HIP_CSRC}" -I "${SLIMT_HIP_INCLUDE}" -o "${SLIMT_HIP_SO}"
    DEPENDS ${SLIMT_HIP_SOURCES} ${SLIMT_HIP_HEADERS} "${SLIMT_HIP_INCLUDE}/slimt_hip.h"
    COMMENT "hipcc: libslimt_hip.so (gfx950)"
    VERBATIM)
  add_custom_target(slimt_hip_build DEPENDS "${SLIMT_HIP_SO}")
  add_library(slimt_hip SHARED IMPORTED GLOBAL)
  set_target_properties(slimt_hip PROPERTIES IMPORTED_LOCATION "${SLIMT_HIP_SO}" IMPORTED_NO_SONAME TRUE)
  target_include_directories(slimt_hip INTERFACE "${SLIMT_HIP_INCLUDE}")
  add_dependencies(slimt_hip slimt_hip_build)
endif()
