# poll delays / pacing of the persistent launch, one at a time around the defaults (tools/ab_options.py; batch 1 and 4)
for sw in "persistent_delay=8,10,12,14" "persistent_delay_act=10,12,14,16" "persistent_delay_x=11,13,15,17" "persistent_delay_att=4,8,12" "persistent_pace=2,3,4"; do
  python tools/ab_options.py fp32 "$sw" --batches ${BATCHES:-1 4} --rounds ${ROUNDS:-3} --steps ${STEPS:-128} 2>/dev/null | cut -c1-20,60-400
done
