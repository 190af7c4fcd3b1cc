import torch, time
d='cuda:0'
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).total_memory>>30)
l=torch.nn.LSTM(300,600,4,batch_first=True).to(d); g=torch.nn.GRU(600,600,2,batch_first=True).to(d)
x=torch.randn(16,30,300,device=d)
with torch.no_grad():
    for i in range(3):
        torch.cuda.synchronize(); t=time.time(); y,_=l(x); z,_=g(y); torch.cuda.synchronize(); print('rnn ms',(time.time()-t)*1e3)
