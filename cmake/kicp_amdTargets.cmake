# kicp_amdTargets.cmake -- the targets a consumer links, under the reference's own names.  Included by the top-level
# CMakeLists.txt (KICP_AMD_ROOT = the source tree) and by kicp_amdConfig.cmake (KICP_AMD_ROOT = the source tree or an install
# prefix).  Reference target graph being replaced (/root/reference/cpp/kinematic_icp):
#   kinematic_icp_pipeline     -> kinematic_icp_registration kinematic_icp_threshold kiss_icp_pipeline   pipeline/CMakeLists.txt:23-27
#   kinematic_icp_registration -> kiss_icp_core Eigen3::Eigen TBB::tbb Sophus::Sophus                   registration/CMakeLists.txt:23-26
#   kinematic_icp_threshold    -> Sophus::Sophus                                                        correspondence_threshold/CMakeLists.txt:23-26
# Here all of them are INTERFACE targets over the drop-in headers; the code behind them is libkicp_amd.so (no TBB: the host
# side has no thread pool of its own).
if(TARGET kinematic_icp_pipeline)
  return()
endif()

if(EXISTS "${KICP_AMD_ROOT}/kinematic_icp_amd/cpp/kicp_bridge.hpp")  # source tree
  set(KICP_AMD_INCLUDE_DIRS "${KICP_AMD_ROOT}/kinematic_icp_amd/cpp" "${KICP_AMD_ROOT}/include")
  set(KICP_AMD_COMPAT_DIR "${KICP_AMD_ROOT}/kinematic_icp_amd/cpp/compat")
  set(KICP_AMD_LIBRARY "${KICP_AMD_ROOT}/kinematic_icp_amd/libkicp_amd.so")
else()  # install prefix
  set(KICP_AMD_INCLUDE_DIRS "${KICP_AMD_ROOT}/include/kicp_amd")
  set(KICP_AMD_COMPAT_DIR "${KICP_AMD_ROOT}/include/kicp_amd/compat")
  find_library(KICP_AMD_LIBRARY kicp_amd HINTS "${KICP_AMD_ROOT}/lib" "${KICP_AMD_ROOT}/lib64" REQUIRED)
endif()

# the C-ABI library (include/kicp.h): imported, built outside CMake's own rules by kinematic_icp_amd/csrc/Makefile
add_library(kicp_amd SHARED IMPORTED GLOBAL)
set_target_properties(kicp_amd PROPERTIES IMPORTED_LOCATION "${KICP_AMD_LIBRARY}" IMPORTED_NO_SONAME TRUE)
get_filename_component(KICP_AMD_LIBDIR "${KICP_AMD_LIBRARY}" DIRECTORY)

# Eigen / Sophus: the real packages where they exist (the drop-in headers compile against them unchanged, kicp_bridge.hpp);
# the stand-ins of cpp/compat otherwise (this image has neither)
if(NOT KICP_AMD_USE_COMPAT_HEADERS)
  find_package(Eigen3 QUIET NO_MODULE)
  find_package(Sophus QUIET NO_MODULE)
endif()

add_library(kicp_amd_headers INTERFACE)
target_compile_features(kicp_amd_headers INTERFACE cxx_std_17)
target_include_directories(kicp_amd_headers INTERFACE ${KICP_AMD_INCLUDE_DIRS})
if(TARGET Eigen3::Eigen AND TARGET Sophus::Sophus AND NOT KICP_AMD_USE_COMPAT_HEADERS)
  target_link_libraries(kicp_amd_headers INTERFACE Eigen3::Eigen Sophus::Sophus)
else()
  target_include_directories(kicp_amd_headers INTERFACE "${KICP_AMD_COMPAT_DIR}")
endif()
# libkicp_amd.so depends on libamdhip64 / libhsa-runtime64 of the ROCm it was built with: let the consumer's link step leave
# those to the loader (the library carries its own DT_NEEDED entries)
target_link_libraries(kicp_amd_headers INTERFACE kicp_amd)
target_link_options(kicp_amd_headers INTERFACE "LINKER:-rpath,${KICP_AMD_LIBDIR}" "LINKER:-rpath,/opt/rocm/lib" "LINKER:--allow-shlib-undefined")

foreach(name kiss_icp_core kiss_icp_pipeline kinematic_icp_threshold kinematic_icp_registration kinematic_icp_pipeline)
  add_library(${name} INTERFACE)
  target_link_libraries(${name} INTERFACE kicp_amd_headers)
endforeach()
# the reference's dependency edges, kept so that `kinematic_icp_pipeline` alone carries everything (ros/CMakeLists.txt:67)
target_link_libraries(kiss_icp_pipeline INTERFACE kiss_icp_core)
target_link_libraries(kinematic_icp_registration INTERFACE kiss_icp_core)
target_link_libraries(kinematic_icp_pipeline INTERFACE kinematic_icp_registration kinematic_icp_threshold kiss_icp_pipeline)
add_library(kicp_amd::kinematic_icp_pipeline ALIAS kinematic_icp_pipeline)
add_library(kicp_amd::kinematic_icp_registration ALIAS kinematic_icp_registration)
add_library(kicp_amd::kinematic_icp_threshold ALIAS kinematic_icp_threshold)
