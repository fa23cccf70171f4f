mkdir -p gpurun_out/r6x
for rep in 1 2 3; do
for v in 0 1 unset; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  echo -n "HIP_FORCE_DEV_KERNARG=$v " >> gpurun_out/r6x/kernarg.txt
  python tools/decode_quick.py --int8 --steps 100 --reps 6 >> gpurun_out/r6x/kernarg.txt 2>&1
  echo -n "HIP_FORCE_DEV_KERNARG=$v " >> gpurun_out/r6x/kernarg.txt
  python tools/decode_quick.py --steps 100 --reps 4 >> gpurun_out/r6x/kernarg.txt 2>&1
done
done
cut -c1-110 gpurun_out/r6x/kernarg.txt
