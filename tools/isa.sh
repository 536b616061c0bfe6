# device ISA of rx_kernels.hip -> /tmp/t/rx.s, the headline kernel alone -> /tmp/t/k_rx4.s, and its register budget
mkdir -p /tmp/t
cd /root/repo/osmo-gmr_amd/csrc && /opt/rocm/bin/hipcc -xhip -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I../../include -I. --offload-arch=gfx950 --cuda-device-only -S rx_kernels.hip -o /tmp/t/rx.s 2>&1 | grep -v "hip-link"
awk '/^_ZN4gmr15k_rx4ILi16ELi4EEEvNS_6RxArgsEiii:/,/s_endpgm/' /tmp/t/rx.s > /tmp/t/k_rx4.s
wc -l /tmp/t/k_rx4.s
for k in k_rx4ILi16ELi4EEEvNS_6RxArgsEiii k_rx4gILi8ELi4EEEvNS_6RxArgsEii k_rx4gILi16ELi4EEEvNS_6RxArgsEii k_rx_chainILi16ELi4ELb0EEEvNS_6RxArgsENS_10RxLoopArgsEiii; do
  echo $k; grep -A40 "\.name:.*$k\$" /tmp/t/rx.s | grep -E "vgpr_count|sgpr_spill|vgpr_spill|group_segment_fixed" | tr '\n' ' '; echo
done
