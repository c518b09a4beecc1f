#!/bin/bash
for v in "" obe1 obe2 obe4 obe8 obe15 oba4; do
  for cfg in "8 3 22 10"; do echo -n "${v:-base}: "; FRT_LIB_VARIANT=$v python tools/exp/fir_only.py $cfg; done
done
