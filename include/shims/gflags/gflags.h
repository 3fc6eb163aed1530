// gflags/gflags.h — the reference's CLI includes this header (cmd/main.cpp:19) but uses nothing
// from it; an empty stand-in keeps that include line compiling.
#pragma once
