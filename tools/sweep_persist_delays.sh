# poll delays / pacing of the persistent launch, one at a time around the defaults (tools/ab_options.py; batch 1 and 4)
for sw in "persistent_delay_att=0,2,4,8" "persistent_delay=8,10,12,14,16" "persistent_delay_act=12,14,16,18,20" "persistent_delay_x=11,13,15,17,19" "persistent_pace=2,3,4" "persistent_nap=0,1,2"; do
  python tools/ab_options.py fp32 "$sw" --batches 1 4 --rounds 3 --steps 128 2>/dev/null | cut -c1-20,60-400
done
