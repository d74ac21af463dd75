// vch_hip.hip — single translation unit of libvch_hip.so (gfx950).
#include "vch_engine2d.hip"
#include "vch_engine1d.hip"
#include "vch_comm.hip"
