cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -X faulthandler -c "
import __graft_entry__ as g
g.smoke()
print('after smoke')
" 2>&1 | tail -30
echo "exit=$?"
