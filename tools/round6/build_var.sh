#!/bin/bash
# build a variant of the device library (and the host library and command against it) into methyldackel_amd/_exp_<name>: build_var.sh name flags...
n=$1; shift; B=methyldackel_amd/_exp_$n; mkdir -p $B
make -s B=$B HIPFLAGS="$*" $B/libmdk_hip.so $B/libmdk_extract.so $B/MethylDackel 2>&1 | grep -E "error|warning: v|spill" ; ls -la $B/libmdk_hip.so | awk '{print $5, $9}'
