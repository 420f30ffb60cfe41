#!/bin/bash
# register / LDS / occupancy table of the kernels of one .hip file whose mangled name matches a pattern
# usage: tools/resusage.sh file.hip pattern
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage -c -o /dev/null "$1" 2>&1 | awk -v pat="$2" '
/Function Name:/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[-R.*/,"",n); name=n}
/ VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)}
/AGPRs:/ {a=$0; sub(/.*AGPRs: /,"",a); sub(/ \[.*/,"",a)}
/ScratchSize/ {s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s)}
/Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)}
/LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); if (name ~ pat) printf "%s vgpr %s agpr %s scratch %s occ %s lds %s\n", name, v, a, s, o, l}' | c++filt | sed 's/void lcx:://; s/(float const[^)]*)//; s/(double const[^)]*)//'
