# kicp_amdConfig.cmake -- `find_package(kicp_amd CONFIG)` for consumers outside this tree (e.g. the reference's ROS package:
# ros/CMakeLists.txt then needs `find_package(kicp_amd REQUIRED)` in place of its add_subdirectory(../cpp/kinematic_icp) and links
# `kinematic_icp_pipeline` as before, line 67).  Works from the source tree (-Dkicp_amd_DIR=<repo>/cmake) and from an install
# prefix (<prefix>/lib/cmake/kicp_amd).
get_filename_component(_kicp_amd_here "${CMAKE_CURRENT_LIST_DIR}" ABSOLUTE)
if(EXISTS "${_kicp_amd_here}/../kinematic_icp_amd/cpp/kicp_bridge.hpp")
  get_filename_component(KICP_AMD_ROOT "${_kicp_amd_here}/.." ABSOLUTE)          # source tree: <repo>/cmake
else()
  get_filename_component(KICP_AMD_ROOT "${_kicp_amd_here}/../../.." ABSOLUTE)    # install prefix: <prefix>/lib/cmake/kicp_amd
endif()
include("${CMAKE_CURRENT_LIST_DIR}/kicp_amdTargets.cmake")
if(NOT EXISTS "${KICP_AMD_LIBRARY}")
  set(kicp_amd_FOUND FALSE)
  set(kicp_amd_NOT_FOUND_MESSAGE "libkicp_amd.so not found at ${KICP_AMD_LIBRARY}: build it with `make -C kinematic_icp_amd/csrc` (hipcc, gfx950)")
else()
  set(kicp_amd_FOUND TRUE)
endif()
