# scratch command file for ad-hoc gpurun experiments
