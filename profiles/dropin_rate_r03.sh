#!/bin/bash
# PCIe-inclusive rates of the drop-in executables (the reference's own main() on the library) at 256^3
R=$PWD
for cfg in "blast step" "blast auto" "ioniz_sphere step" "ioniz_sphere auto"; do
  set -- $cfg
  d=$(mktemp -d)
  ( cd $d && AA_COHERENCE=$2 timeout -k 10 300 $R/oracle/_ref/athena_$1_amd -i $R/atmospheric-athena_amd/decks/athinput.$1 -d $d/run domain1/Nx1=256 domain1/Nx2=256 domain1/Nx3=256 time/nlim=40 > out.txt 2> err.txt; echo "$1 coherence=$2: $(grep 'zone-cycles/wall-second' out.txt | tail -1)  $(grep -c 'Radiation done' err.txt) ion steps" )
  rm -rf $d
done
# and two slabs on the one device (rehearsal of AA_NGPU)
d=$(mktemp -d); ( cd $d && AA_NGPU=2 timeout -k 10 300 $R/oracle/_ref/athena_blast_amd -i $R/atmospheric-athena_amd/decks/athinput.blast -d $d/run domain1/Nx1=256 domain1/Nx2=256 domain1/Nx3=256 time/nlim=40 > out.txt 2> err.txt; echo "blast AA_NGPU=2 (one device): $(grep 'zone-cycles/wall-second' out.txt | tail -1)" ); rm -rf $d
