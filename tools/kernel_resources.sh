#!/bin/bash
# Registers, LDS and occupancy of the kernels whose name contains $1, from the build's ISA listing.
f=$(dirname "$0")/../particle-simulation_amd/build/paircount_gfx950.s
awk -v pat="$1" '/^\t\.amdhsa_kernel /{name=$2} /^; (NumVgprs|NumSgprs|ScratchSize|LDSByteSize|Occupancy)/{ if (index(name, pat)) printf "%s %s\n", name, $0 }' "$f" 
