# cmake -DIN=<reference>/src/scalable_ccd/config.hpp.in -DOUT=<oracle/_ref/include/scalable_ccd/config.hpp> -P ref_config.cmake
# The reference's own configure_file() (CMakeLists.txt:105-107) run in script mode on the reference's own template, with the
# project() values of CMakeLists.txt:63-66 and the option defaults of :68-72 -- the CPU build: no CUDA, double precision.
set(PROJECT_NAME "ScalableCCD")
set(PROJECT_VERSION "0.1.0")
set(PROJECT_VERSION_MAJOR "0")
set(PROJECT_VERSION_MINOR "1")
set(PROJECT_VERSION_PATCH "0")
set(SCALABLE_CCD_WITH_CUDA OFF)
set(SCALABLE_CCD_USE_DOUBLE ON)
set(SCALABLE_CCD_TOI_PER_QUERY OFF)
set(SCALABLE_CCD_WITH_PROFILER OFF)
configure_file("${IN}" "${OUT}")
