/* placeholder: replaced by the T.800 restatement */ int qo_j2k_placeholder(void){return 0;}
