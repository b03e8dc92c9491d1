#!/bin/bash
# What the GPU box's sysfs offers cartnet_amd/telemetry.py (run once per pool image; output under gpurun_out/).
for d in /sys/class/drm/card*/device; do
  echo "== $d -> $(readlink -f $d)"
  ls $d | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk; do echo "-- $f"; cat $d/$f 2>&1 | head -5; done
  for h in $d/hwmon/hwmon*; do
    echo "-- $h"; ls $h | tr '\n' ' '; echo
    for f in $h/power1_average $h/power1_input $h/power1_cap $h/temp*_input $h/temp*_label $h/freq*_input $h/freq*_label; do
      [ -e $f ] && echo "$(basename $f) = $(cat $f 2>&1)"
    done
  done
done
python3 -c "
import torch
from cartnet_amd import telemetry as t
print(t.pci_address(0)); print(t.read(0))"
