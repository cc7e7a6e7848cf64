# placement probing: 4 fresh processes per setting, stand-alone sweep ms and q/s
for t in 1 3 5; do for i in 1 2 3 4; do echo "== db_place_tries=$t run $i"; SPIRAL_ALLOC_DEBUG=1 SPIRAL_DB_PLACE_TRIES=$t STEPS=10 python scripts/r03_ab_switches.py 2>&1 | grep -E "placement|baseline" | head -2; done; done
