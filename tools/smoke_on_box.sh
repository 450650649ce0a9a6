export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
time python - <<'PY'
import __graft_entry__ as g
g.smoke()
print("smoke ok")
PY
