import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d=json.loads(l)
print("value",round(d["value"]), [round(x) for x in d["value_repeats"]], "single",round(d["single_frame_iters_per_s"]), "plain",round(d["plain_loop_iters_per_s"]), "python",round(d["python_loop_iters_per_s"]), "stream",round(d.get("stream_of_frames_iters_per_s", 0)))
print("single", d["kernels_ms_per_iter_native_single_frame"])
print("plain ", d["kernels_ms_per_iter_native_plain_loop"])
if "train_step" in d:
    for r in d["train_step"]["per_P"]: print(r)
