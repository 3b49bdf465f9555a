# shared by the evidence scripts: run one GPU step under its own timeout, log it, and END the script when the step was
# killed (timeout 124 / 137) or died of a signal / abort (rc >= 128, e.g. 134 after a GPU fault): no further GPU step
# is started after such a step.  An ordinary non-zero exit is reported and returned to the caller.
run() { # name, timeout, command...
  local name=$1 t=$2; shift 2
  timeout -k 10 $t "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "step $name was killed or aborted (rc=$rc): stopping, no further GPU step"; exit 1; fi
  return $rc
}
