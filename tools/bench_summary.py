import json,sys
d=json.load(open(sys.argv[1])); print(round(d["value"],1), round(d["value_resident"],1), round(d["value_streamed"] or 0,1), round(d["single_stream"]["value"],1), round(d["single_caller"]["value"],1), (d.get("scale_reference") or {}).get("batches_in_flight_per_gpu"))
